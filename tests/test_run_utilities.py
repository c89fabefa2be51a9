"""Run utilities and each() (reference: src/edt.pyx:847-994, src/edt_voxel_graph.hpp:238-310) against
golden vectors recorded from the reference module (tests/golden/make_golden_runs.py).  These are host-side
helpers around the DT: no GPU needed for the first group; the device-resident each() is a `gpu` test."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT

# EDT_TEST_MODULE_DIR: run the same checks against another build of the `edt` module -- the reference's
# unmodified Cython binding compiled against our drop-in headers (tests/test_cython_dropin.py)
sys.path.insert(0, os.environ.get("EDT_TEST_MODULE_DIR") or os.path.join(ROOT, "euclidean-distance-transform-3d_amd"))

GOLD = np.load(os.path.join(ROOT, "tests", "golden", "edt_runs.npz"))
NCASES = int(GOLD["ncases"])


def case(t):
    pre = f"{t:03d}/"
    return {k[len(pre):]: GOLD[k] for k in GOLD.files if k.startswith(pre)}


def runs_of(c):
    out, at = {}, 0
    for k, n in zip(c["keys"].tolist(), c["counts"].tolist()):
        out[k] = [tuple(p) for p in c["pairs"][at:at + n].tolist()]
        at += n
    return out


@pytest.mark.parametrize("t", range(NCASES))
def test_runs_draw_transfer_erase_match_the_reference(t):
    import edt
    c = case(t)
    want = runs_of(c)
    got = edt.runs(c["labels"])
    assert list(got.keys()) == list(want.keys())          # std::map order
    assert {k: [tuple(p) for p in v] for k, v in got.items()} == want
    k0 = int(c["draw_key"])
    assert np.array_equal(edt.draw(7, got[k0], np.zeros_like(c["labels"])), c["draw"])
    dest = np.full(c["labels"].shape, -1.0, dtype=np.float32, order="F" if t % 2 else "C")
    assert np.array_equal(edt.transfer(got[k0], c["dt"], dest), c["transfer"])
    assert np.array_equal(edt.erase(got[k0], c["dt"].copy(order="K")), c["erase"])


@pytest.mark.parametrize("t", range(NCASES))
@pytest.mark.parametrize("in_place", [False, True])
def test_each_matches_the_reference(t, in_place):
    import edt
    c = case(t)
    it = edt.each(c["labels"], c["dt"], in_place=in_place)
    keys, imgs = c[f"each{int(in_place)}_keys"], c[f"each{int(in_place)}_imgs"]
    assert len(it) == len(keys)
    seen = 0
    for (k, img), wk, wimg in zip(it, keys.tolist(), imgs):
        assert k == wk and np.array_equal(img, wimg)
        assert img.flags.writeable != in_place
        seen += 1
    assert seen == len(keys)


def test_invalid_runs_raise():
    import edt
    img = np.zeros(10, dtype=np.float32)
    for bad in ([(-1, 3)], [(0, 11)], [(4, 4)], [(5, 2)]):
        with pytest.raises(RuntimeError, match="Invalid run"):
            edt.draw(1, bad, img)
        with pytest.raises(RuntimeError, match="Invalid run"):
            edt.transfer(bad, img, img.copy())
    if not os.environ.get("EDT_TEST_MODULE_DIR"):
        # (the reference's own Python layer indexes labels[0] of an empty array and raises IndexError,
        # src/edt.pyx:894; this repo's module returns the empty map extract_runs gives for 0 voxels)
        assert edt.runs(np.zeros((0,), dtype=np.uint8)) == {}


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["uint8", "int16", "int32", "int64", "float32"])
@pytest.mark.parametrize("in_place", [False, True])
def test_device_each_equals_the_reference_definition(dtype, in_place):
    """edt.device.each: dt * (labels == label) for every label, computed on the device."""
    import torch
    from edt import device
    rng = np.random.default_rng(3)
    shape = (37, 41, 45)
    lab = rng.integers(-3 if dtype.startswith("int") else 0, 9, size=shape).astype(dtype)
    lab[rng.random(shape) < 0.2] = 0
    tl = torch.from_numpy(lab).cuda()
    dt = device.edt(tl, anisotropy=(3, 2, 1))
    mdt = dt.cpu().numpy()
    keys = [k for k in np.unique(lab).tolist() if k != 0]
    it = device.each(tl, dt, in_place=in_place)
    assert len(it) == len(keys)
    got_keys = []
    for k, img in it:
        got_keys.append(k)
        assert np.array_equal(img.cpu().numpy(), (lab == k) * mdt)
    assert sorted(got_keys) == sorted(keys)


def test_module_level_helpers_reshape_and_nvl():
    """`edt.reshape` / `edt.nvl` (src/edt.pyx:115-118, :851-877): views of contiguous arrays under another shape in their own
    memory order; the default for a missing argument."""
    import edt
    assert edt.nvl(None, 3) == 3 and edt.nvl(0, 3) == 0
    a = np.arange(24, dtype=np.int32)
    f = edt.reshape(np.asfortranarray(a.reshape(4, 6)), (2, 12))
    assert f.flags.f_contiguous and np.array_equal(f, np.asfortranarray(a.reshape(4, 6)).reshape((2, 12), order="F"))
    c = edt.reshape(a.reshape(4, 6), (2, 3, 4))
    assert c.flags.c_contiguous and np.shares_memory(c, a) and np.array_equal(c, a.reshape(2, 3, 4))
    assert np.shares_memory(edt.reshape(a, (6, 4), order="F"), a)                 # 1-D: contiguous both ways, still a view
    assert np.array_equal(edt.reshape(a.reshape(4, 6)[:, ::2], (6, 2)), a.reshape(4, 6)[:, ::2].reshape(6, 2))
