"""Build-time guard of the wide-state kernels (csrc/wide.hip): 1024-lane workgroups cap every lane at 128 VGPRs, so these kernels live
at the cap with a few hundred bytes of scratch per lane and no headroom.  Round 1 met a wrong-result event when an unrelated fp64
kernel added to the translation unit changed the code generated for the untouched fp64 sampler (dead end 11): the unit
then had out-of-line device functions (gj_solve, gauss2) whose inlining the compiler re-decided.  Every device helper of wide.hip is
now __forceinline__ -- a kernel's code no longer depends on what else the unit contains -- and this test keeps it so:
  * the unit emits no device function besides its kernels (nothing is called out of line);
  * no kernel's scratch grows past the committed table (profiles/r04_wide_resources.json; round 3's is r03_wide_resources.json: the round-4 column kernels of
    wide_shared.h were added to the unit and the table regenerated deliberately -- every kernel of round 3 kept its registers and scratch to the byte) by more than
    64 bytes per lane;
  * nothing exceeds the 128-register cap or reports a dynamic stack.
Compiles wide.hip once with -Rpass-analysis=kernel-resource-usage (CPU only, about half a minute)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_wide_kernels_are_self_contained_and_do_not_spill_more():
    import kernel_resources as KR
    now = KR.table("wide.hip")
    want = json.load(open(os.path.join(ROOT, "profiles", "r04_wide_resources.json")))
    assert now, "no resource remarks: is hipcc there?"
    stray = [k for k in now if "wk_" not in k]
    assert not stray, f"device functions emitted out of line in wide.hip: {stray}"
    assert set(now) == set(want), sorted(set(now) ^ set(want))
    for k, v in now.items():
        assert v.get("VGPRs", 0) + v.get("AGPRs", 0) <= 128, (k, v)
        grow = v.get("ScratchSize [bytes/lane]", 0) - want[k].get("ScratchSize [bytes/lane]", 0)
        assert grow <= 64, f"{k}: scratch grew by {grow} B/lane over profiles/r04_wide_resources.json (regenerate it deliberately if intended)"
