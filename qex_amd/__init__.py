"""qex_amd -- MI355X-native staggered Dslash + CG (+ Wilson flow) behind QEX's operator API.

The product is qex_amd/libqexhip.so (hand-written HIP for gfx950 + RCCL) with the C ABI of
include/qexhip.h; this package is the thin host-side mirror of the reference interface used by
the tests and the benchmark.  There is no CPU fallback.
"""
from ._lib import QexHipError, LIB_PATH, lib  # noqa: F401
from .layout import Layout  # noqa: F401
from .gauge import setBC, stagPhase, rephase, unit, synthetic_random_su3, synthetic_gaussian_vector  # noqa: F401
from .staggered import (  # noqa: F401
    Context, device_count, Staggered, SolverParams, newStag, newStag3, plaq, gaugeForce, gaugeFlow, gaugeSet, gaugeFlowResident, flowEQ, flowMeasure, gaugeAction, gaugeUpdate, reunit, wline, ploops, s4_gauge, ResidentMD, HisqCoefs, HypCoefs, makeImpLinks, fat7lDeriv, EVEN, ODD, ALL,
)
from .io import loadGauge, loadGaugeSlab, saveGauge, getFileLattice, gaugeFileInfo, writeField, readField, fileMetadata  # noqa: F401
from .rng import RngField, RngMilc6, MRG32k3a  # noqa: F401
