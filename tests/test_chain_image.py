"""The chain image (include/pgbart_image.h, pgb_checkpoint_*): CPU half -- the oracle writes and reads the record
every backend shares; the layout is pinned from Python, and damaged records are errors.  The GPU half (the HIP
library writes the same bytes, chains migrate between backends, the oracle resumes a GPU chain after its burn-in)
is tests/test_steady_state_gpu.py."""
import ctypes as C

import numpy as np
import pytest

from _cases import digest, make_case, run_case
from pymc_bart_amd import _abi
from pymc_bart_amd.image import ChainImage, ImageHeader, differing_fields
from pymc_bart_amd.sampler import PyBartSettings, PySampler

ROUND_TRIP = ["nan_onehot_prior", "categorical_k4_cfg5_small", "linear_response", "categorical_k3_mix", "subset_rule",
              "probit_cfg4_small", "upstream/onehot_fail_nan", "one_tree_two_particles"]


@pytest.mark.parametrize("name", ROUND_TRIP)
def test_oracle_chain_resumed_from_its_image_is_the_uninterrupted_chain(oracle, name):
    c = make_case(name)
    cuts = (1, c["steps"] // 2 - 1, c["steps"] // 2 + 2)
    assert digest(run_case(c, oracle, checkpoint_at=cuts)) == digest(run_case(c, oracle))


def _sampler(c, backend, seed=None):
    X, Y = c["X"], c["Y"]
    p = X.shape[1]
    st = PyBartSettings.from_data(X, c.get("bart_Y", Y), m=c["m"], num_particles=c["P"], seed=c["seed"] if seed is None else seed,
                                  batch=c["batch"], family=c.get("family", "normal"), n_outputs=c.get("K", 1),
                                  response=c.get("response", "constant"), compat=c.get("compat", 0))
    rules = np.zeros(p, np.int32) if c["rules"] is None else c["rules"]
    prior = np.ones(p) if c["prior"] is None else c["prior"]
    s = PySampler(st, X, Y, rules, prior, backend=backend)
    s.set_likelihood([0.8] if c.get("family", "normal") == "normal" else c.get("lik_params", []))
    return s


def test_image_layout_as_documented(oracle):
    """The record parsed by an independent reader (pymc_bart_amd/image.py follows the header's text, not its code):
    header fields, section sizes, and the content against what the ABI's own getters report."""
    c = make_case("categorical_k3_mix")
    s = _sampler(c, oracle)
    for it in range(9):
        s.step(it < 6)
    blob = s.checkpoint()
    img = ChainImage.parse(blob)
    hd = img.header
    assert hd.header_bytes == C.sizeof(ImageHeader) and hd.total_bytes == len(blob) and img.writer == "oracle-cpu"
    assert bytes(hd.s) == bytes(s.settings.as_c())
    stt = s.state()
    assert hd.iter == stt["iter"] and hd.lower == stt["lower"] and np.array_equal(img.leaf_sd, stt["leaf_sd"])
    assert hd.rs_count == 6 * s.settings.batch_sizes()[0]
    n, K, m = s.settings.n, 3, s.settings.m
    assert img.sum_trees.shape == (K, n) and img.lid.shape == (m, n)
    forest = s.export_trees(1)
    assert np.array_equal(img.node_off, forest.node_off) and np.array_equal(img.var, forest.var)
    assert np.array_equal(img.count, forest.count) and np.array_equal(img.value, forest.value)
    assert np.array_equal(img.slope, forest.slope) and np.array_equal(img.svar, forest.svar)
    assert np.array_equal(img.left, forest.left) and np.array_equal(img.right, forest.right)
    # the labels say which leaf every row sits in: the row counts of the leaves, and sum_trees itself
    base = np.repeat(img.node_off[:-1], np.diff(img.node_off))
    for t in (0, m - 1):
        sl = slice(img.node_off[t], img.node_off[t + 1])
        leaves = np.flatnonzero(img.var[sl] < 0)
        for k in leaves:
            assert np.count_nonzero(img.lid[t] == img.label[sl][k]) == img.count[sl][k]
    assert base.size == hd.total_nodes
    # split weights: alpha in the units of the caller's prior
    w = s.split_weights()
    assert np.allclose(img.alpha / img.alpha.sum(), w / w.sum(), rtol=1e-12)
    assert np.array_equal(np.cumsum(img.alpha)[-1:] >= img.cdf[-1:], [True])  # the sampler's sums may lag the weights
    # a second image of the same state is the same bytes; one step later it is not the same chain state
    assert s.checkpoint() == blob
    s.step(False)
    assert "sum_trees" in differing_fields(ChainImage.parse(s.checkpoint()), img)


def test_loading_an_image_then_saving_gives_the_same_bytes(oracle):
    c = make_case("nan_onehot_prior")
    a = _sampler(c, oracle)
    for it in range(7):
        a.step(it < 4)
    blob = a.checkpoint()
    b = _sampler(c, oracle)
    b.restore(blob)
    assert b.checkpoint() == blob
    sa, va = a.step(False)
    sb, vb = b.step(False)
    assert np.array_equal(sa, sb) and np.array_equal(va, vb)
    assert a.counters.as_dict() == b.counters.as_dict()
    ta, tb = a.export_trees(0), b.export_trees(0)
    for f in ("tree_id", "node_off", "var", "count", "split", "value"):
        assert np.array_equal(getattr(ta, f), getattr(tb, f)), f


def test_export_of_the_last_batch_survives_the_image(oracle):
    c = make_case("subset_rule")
    a = _sampler(c, oracle)
    for it in range(5):
        a.step(it < 3)
    b = _sampler(c, oracle)
    b.restore(a.checkpoint())
    ta, tb = a.export_trees(0), b.export_trees(0)
    for f in ("tree_id", "node_off", "var", "left", "right", "count", "split", "value", "rule"):
        assert np.array_equal(getattr(ta, f), getattr(tb, f)), f


def test_damaged_or_foreign_images_are_refused_with_a_message(oracle):
    c = make_case("ragged_1025")
    a = _sampler(c, oracle)
    for it in range(4):
        a.step(True)
    blob = a.checkpoint()
    b = _sampler(c, oracle)

    def refused(bad, text):
        with pytest.raises(_abi.PGBError, match=text):
            b.restore(bytes(bad))

    refused(blob[:100], "truncated")
    refused(blob[:-8], "truncated")
    refused(b"XXXXXXXX" + blob[8:], "not a pgbart checkpoint")
    bad = bytearray(blob)
    bad[8] = 99  # version
    refused(bad, "layout version")
    other = _sampler(c, oracle, seed=1)  # another chain of the same model: settings differ in the seed
    with pytest.raises(_abi.PGBError, match="settings differ"):
        other.restore(blob)
    img = ChainImage.parse(blob)
    off = img.node_off.ctypes.data - np.frombuffer(blob, np.uint8).ctypes.data
    bad = bytearray(blob)
    bad[off + 4: off + 8] = (10_000).to_bytes(4, "little")  # node_off[1]
    refused(bad, "inconsistent")
    off = img.var.ctypes.data - np.frombuffer(blob, np.uint8).ctypes.data
    bad = bytearray(blob)
    bad[off: off + 4] = (77).to_bytes(4, "little")  # a split on a column X does not have
    refused(bad, "inconsistent")
    # the sampler that refused all of these is untouched: it still loads the good image and continues the chain
    b.restore(blob)
    sa, _ = a.step(False)
    sb, _ = b.step(False)
    assert np.array_equal(sa, sb)


def test_binding_refuses_a_library_of_another_abi_revision(tmp_path):
    """pymc_bart_amd/_abi.py checks pgb_abi_version() (include/pgbart.h: PGB_ABI_VERSION) before binding anything."""
    import subprocess

    src = tmp_path / "stale.c"
    src.write_text("int pgb_abi_version(void) { return 5; }\nconst char* pgb_backend_name(void) { return \"hip-gfx950\"; }\n")
    so = tmp_path / "libstale.so"
    subprocess.check_call(["gcc", "-shared", "-fPIC", str(src), "-o", str(so)])
    with pytest.raises(_abi.PGBError, match="revision 5"):
        _abi.PGBLibrary(str(so))
    src.write_text("int pgb_create(void) { return 0; }\n")
    so2 = tmp_path / "libolder.so"  # (another name: the loader caches by path)
    subprocess.check_call(["gcc", "-shared", "-fPIC", str(src), "-o", str(so2)])
    with pytest.raises(_abi.PGBError, match="no pgb_abi_version"):
        _abi.PGBLibrary(str(so2))


def test_the_forest_of_an_image_predicts_like_the_sampler_it_came_from(oracle):
    """`ChainImage.forest()` = `export_trees(1)` of the chain: a checkpoint can be predicted from without a sampler,
    and its in-sample prediction is the image's own sum_trees (one-hot / subset splits included: the caller names the
    columns' rules, the image holds trees)."""
    from pymc_bart_amd.trees import PosteriorSampler

    c = make_case("subset_rule")
    s = _sampler(c, oracle)
    for it in range(8):
        s.step(it < 4)
    img = ChainImage.parse(s.checkpoint())
    f_img, f_smp = img.forest(rules=c["rules"]), s.export_trees(1)
    for name in ("tree_id", "node_off", "var", "left", "right", "count", "split", "value", "rule"):
        assert np.array_equal(getattr(f_img, name), getattr(f_smp, name)), name
    m = c["m"]
    ps = PosteriorSampler(f_img, np.arange(m, dtype=np.int32)[None, :], m, 1, backend=oracle)
    # rows without a missing value in a split column predict exactly what the chain holds for them
    X = c["X"]
    ok = ~np.isnan(X).any(axis=1)
    pred = ps.sample_posterior(X[ok], [0])[0, 0]
    np.testing.assert_allclose(pred, img.sum_trees[0][ok], rtol=0, atol=1e-9)
