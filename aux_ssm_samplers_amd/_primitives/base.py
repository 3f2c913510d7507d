"""Boundary types (reference: aux_samplers/_primitives/base.py:8-10)."""
from dataclasses import dataclass
from typing import Any


@dataclass
class SamplerState:
    x: Any
