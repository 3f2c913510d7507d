"""Copies the DATA files of the reference's Lorenz example into fixtures (data, not code): the 201 observation rows `t, y2, y3` of
examples/lorenz/data.csv (its header says x1,x2,x3) and the rows of examples/lorenz/true_xs.csv (`t, x1, x2, x3`, dt = 2e-4) at the observation
times.  Run in the build container, where /root/reference exists:  python tests/golden/make_lorenz_fixture.py"""
import os

import numpy as np

REF = "/root/reference/aux_samplers/examples/lorenz"
HERE = os.path.dirname(os.path.abspath(__file__))
data = np.loadtxt(os.path.join(REF, "data.csv"), delimiter=",", skiprows=1)
true_xs = np.loadtxt(os.path.join(REF, "true_xs.csv"), delimiter=",", skiprows=1)
every = int(round((data[1, 0] - data[0, 0]) / (true_xs[1, 0] - true_xs[0, 0])))
at_obs = true_xs[::every]
assert at_obs.shape[0] == data.shape[0] and np.allclose(at_obs[:, 0], data[:, 0])
np.savetxt(os.path.join(HERE, "lorenz_data.csv"), data, delimiter=",", header="t,y2,y3", comments="", fmt="%.17g")
np.savetxt(os.path.join(HERE, "lorenz_true_xs_at_obs.csv"), at_obs, delimiter=",", header="t,x1,x2,x3", comments="", fmt="%.17g")
print(data.shape, at_obs.shape)
