"""Committed golden fixtures (tests/golden/, produced by tests/golden/make_golden.py from this
repository's oracle -- the reference has no numeric vectors for this path)."""
import json
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
TINY = os.path.join(HERE, "golden", "tiny_n64.npz")
SMALL = os.path.join(HERE, "golden", "small_n4096.json")
sys.path.insert(0, os.path.join(HERE, "golden"))


def test_oracle_reproduces_tiny_vectors(ob):
    g = np.load(TINY)
    o = ob.Oracle(64, 2, 65537)
    assert (o.moduli == g["moduli"]).all()
    assert (o.ntt(0, g["ntt_in"]) == g["ntt_out"]).all()
    assert (o.keygen(11) == g["sk"]).all()
    assert (o.mul(g["idx"][0, 0], g["idx"][1, 0], g["evk"]) == g["mul"]).all()
    res = o.pie_run(g["idx"], g["minus"], g["db"], g["masks"], g["evk"])
    assert (res == g["res"]).all()
    dec = np.stack([o.decrypt_slots(g["sk"], res[bn], 8)[0] for bn in range(res.shape[0])])
    assert (dec == g["dec"]).all()
    assert (np.sort(ob.client_scan(g["ctab"], dec)) == g["found"]).all()
    db = np.stack([o.encode_eval(s) for s in g["slots"].reshape(-1, 8)]).reshape(g["db"].shape)
    assert (db == g["db"]).all()


def test_oracle_reproduces_small_digests(ob):
    import make_golden
    meta = json.load(open(SMALL))
    c = make_golden.build_case(**meta["params"])
    assert [int(v) for v in c["o"].moduli] == meta["moduli"]
    for k, want in meta["digests"].items():
        assert make_golden.sha(c[k]) == want, k
    assert [int(v) for v in c["found"]] == meta["found"]


@pytest.mark.gpu
def test_gpu_reproduces_golden():
    """the HIP path against the committed vectors (tiny ring through the radix-2 kernel, N=4096 through
    the register-blocked kernel)"""
    import make_golden
    from nested_hashing_psi_amd import pie
    g = np.load(TINY)
    cc = pie.PieContext(64, 2, 65537)
    cc.load_relin_key(g["evk"])
    assert (cc.ntt(g["ntt_in"].reshape(1, -1), 0, 1)[0] == g["ntt_out"]).all()
    assert (cc.EvalMult(g["idx"][0, 0], g["idx"][1, 0]) == g["mul"]).all()
    for kw in (dict(vectorizedHCT=g["db"], preCalcRandomMask=g["masks"]), dict(slots=g["slots"], mask_slots=g["mask_slots"])):
        op = pie.BatchedFHEHIPPIE(cc, **kw)
        op.setMinusCompareElement(g["minus"])
        op.setIndex(g["idx"])
        op.run()
        assert (op.getResultList() == g["res"]).all()
    cc.close()
    meta = json.load(open(SMALL))
    c = make_golden.build_case(**meta["params"])
    p = meta["params"]
    cc = pie.PieContext(p["N"], p["L"], p["t"])
    cc.load_relin_key(c["evk"])
    assert make_golden.sha(cc.ntt(c["ntt_in"].reshape(1, -1), 0, 1)[0]) == meta["digests"]["ntt_out"]
    op = pie.BatchedFHEHIPPIE(cc, slots=c["slots"], mask_slots=c["mask_slots"])
    op.setMinusCompareElement(c["minus"])
    op.setIndex(c["idx"])
    op.run()
    assert make_golden.sha(op.getResultList()) == meta["digests"]["res"]
    cc.close()
