"""CPU: the generation fixture (reference's own LinearTransformer(is_training=False) + its numpy samplers,
tests/golden/make_golden.py::dqn_generation_small) pins (1) the oracle's recurrent model and (2) the host-side
samplers of the product (`sampling.sample_cw`: same draws from a seeded np.random, same order)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
from fill import fill_params  # noqa: E402

import rlmg_amd  # noqa: E402,F401
from rlmg_amd.sampling import sample_cw  # noqa: E402
from oracle import cw_model  # noqa: E402

FIX = np.load(os.path.join(HERE, "golden", "dqn_generation_small.npz"))


def _split(row, n_class):
    outs, o = [], 0
    for n in n_class:
        outs.append(row[o:o + n])
        o += n
    return outs


def test_samplers_reproduce_reference_token_stream():
    """Feeding the recorded logits to sample_cw under the recorded np seed gives the recorded tokens."""
    n_class = [int(v) for v in FIX["n_class"]]
    np.random.seed(int(FIX["np_seed"]))
    for t, row in enumerate(FIX["logits"]):
        got = sample_cw(_split(row, n_class))
        assert got.tolist() == FIX["tokens"][t + 1].tolist(), t


def test_oracle_recurrent_generation_matches_fixture():
    n_class = [int(v) for v in FIX["n_class"]]
    ref = fill_params(cw_model.CWLinearTransformer(n_class, 128, 2, 2, variant="dqn", recurrent=True),
                      seed=int(FIX["fill_seed"])).eval()
    np.random.seed(int(FIX["np_seed"]))
    tok = FIX["tokens"][0]
    mem = None
    with torch.no_grad():
        for t in range(len(FIX["logits"])):
            h, mem = ref.forward_hidden(torch.from_numpy(tok).long().view(1, 1, 6), memory=mem, is_training=False)
            assert np.abs(h.numpy().reshape(-1) - FIX["h"][t]).max() < 1e-5
            ys = [y.numpy().reshape(-1) for y in ref.forward_output(h)]
            assert np.abs(np.concatenate(ys) - FIX["logits"][t]).max() < 1e-5
            tok = sample_cw(ys)
            assert tok.tolist() == FIX["tokens"][t + 1].tolist(), t
