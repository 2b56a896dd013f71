"""Pins the numpy restatement of the frozen RAFT flow network (oracle/np_raft.py) against vectors produced by the
reference's RAFT_core (tests/golden/make_golden.py: case_raft).  CPU only."""
import numpy as np

from oracle import np_raft
from oracle.det_init import det_tensor

from helpers import golden, raft_images, raft_state


def test_raft_oracle_matches_reference_vectors():
    fx = golden("raft_basic")
    sd = raft_state(fx)
    a, b = raft_images("raft_basic", (1, 3, 128, 192))
    tr = {}
    low, up = np_raft.raft_forward(sd, a, b, iters=4, trace=tr)
    assert np.abs(tr["fmap1"] - fx["fmap1"]).max() < 2e-4
    assert np.abs(tr["cnet"] - fx["cnet"]).max() < 5e-4
    assert np.abs(tr["flows"][0] - fx["flow_low_it1"]).max() < 2e-4
    assert np.abs(low - fx["flow_low_it4"]).max() < 1e-3       # flows reach |13.7| px at 1/8 resolution
    assert np.abs(up - fx["flow_up_it4"]).max() < 4e-3         # x8
    # the lookup alone, at non-integer positions (corr.py:31-52, incl. the x<-dy / y<-dx window quirk)
    f = np_raft.basic_encoder(np.concatenate([2 * (a / np.float32(255)) - 1, 2 * (b / np.float32(255)) - 1], 0), sd,
                              "fnet", "instance")
    c0 = np_raft.CorrBlock(f[:1], f[1:])(np_raft.coords_grid(1, 16, 24, np.float32) + np.float32(0.37))
    assert np.abs(c0 - fx["corr0"]).max() < 5e-4 * max(1.0, np.abs(fx["corr0"]).max())


def test_raft_shared_norm_keys():
    """norm3 and downsample.1 are one module in the reference: both key sets exist, the later one is what runs."""
    fx = golden("raft_basic")
    keys = [str(k) for k in fx["sd_keys"]]
    assert "cnet.layer2.0.norm3.weight" in keys and "cnet.layer2.0.downsample.1.weight" in keys
    assert not np.array_equal(det_tensor("cnet.layer2.0.norm3.weight", (96,)),
                              det_tensor("cnet.layer2.0.downsample.1.weight", (96,)))
