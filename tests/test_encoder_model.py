"""The structure k_fse_encode_wave's repair rounds rest on (EXPERIMENTS section 2), checked on the CPU with the oracle's tables:
for a fixed run of symbols the map start state -> (bits emitted, end state) of FSE_encodeSymbol (lib/fse.h:514-521) is a monotone step
function with few values, and keeping one older sample per lane never costs a round (scripts/sim/repair_policies.py is the model)."""
import importlib.util
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_spec = importlib.util.spec_from_file_location("repair_policies", os.path.join(ROOT, "scripts", "sim", "repair_policies.py"))
model = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(model)


@pytest.mark.parametrize("proba", [80, 14])
def test_lifted_map_is_a_monotone_step_function_and_the_kept_sample_never_costs_a_round(proba):
    from oracle.oracle import Oracle
    orc = Oracle()
    blocks = orc.probagen_batch(proba, 2)
    for b in range(2):
        T, chains = model.block_maps(orc, np.asarray(blocks[b]), 2.0)
        for E, LE, LM, g in chains:
            assert len(E) == 16
            for k in range(len(E)):
                assert np.all(np.diff(LE[k]) >= 0) and np.all(np.diff(LM[k]) >= 0)        # monotone in the start state
                assert len(np.unique(LE[k])) <= 16                                           # ... with a handful of values
                assert LE[k][-1] - LE[k][0] <= 4 * T                                         # one turn of the circle = at most one bit more
        cur = model.rounds(T, chains, 1, False, False)
        kept = model.rounds(T, chains, 2, False, False)
        assert kept <= cur
        if proba == 14:
            assert cur <= 2 and all(len(np.unique(E[k])) == 1 for E, _, _, _ in chains for k in range(1, len(E)))   # fast mixing: the end forgets the start
