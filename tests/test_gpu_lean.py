"""A sample set declared lean (admm_csr_set_lean) frees its row-major forms once its copies exist: its
whole-set sweeps give the same bits as the full set's, it holds far fewer bytes, and whatever would read rows says
so instead of reading freed memory."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from admm_amd import capi
    c = capi.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("explicit", [False, True])
@pytest.mark.parametrize("order", ["relabel-first", "copy-first", "declare-last"])
def test_lean_set_sweeps_like_the_full_one(ctx, explicit, order):
    from admm_amd import capi
    rows, dim = 60_000, 1 << 16
    rng = np.random.default_rng(5)
    x = ctx.vec(dim, rng.normal(0, 0.1, dim).astype(np.float32))
    z = ctx.vec(dim, rng.normal(0, 0.1, dim).astype(np.float32))

    def build(lean):
        s = ctx.csr_synth(rows, dim, 40.0, seed=3, explicit_values=explicit)
        if lean and order != "declare-last":
            s.set_lean()
        if order == "copy-first":
            s.build_transposed(col_block=4096).relabel_by_frequency()
        else:
            s.relabel_by_frequency().build_transposed(col_block=4096)
        if lean and order == "declare-last":
            s.set_lean()
        return s

    full, lean = build(False), build(True)
    g0, g1 = ctx.vec(dim), ctx.vec(dim)
    f0 = full.logistic_fg(x, g0, z=z, rho=0.05)
    f1 = lean.logistic_fg(x, g1, z=z, rho=0.05)
    assert f0 == f1
    np.testing.assert_array_equal(g0.download(), g1.download())
    np.testing.assert_array_equal(full.predict([x]), lean.predict([x]))
    assert full.metrics([x]) == lean.metrics([x])
    xs0, xs1 = ctx.vec(dim), ctx.vec(dim)
    r0 = full.lbfgs_solve(xs0, z=z, rho=0.05, max_iter=3)
    r1 = lean.lbfgs_solve(xs1, z=z, rho=0.05, max_iter=3)
    assert r0 == r1
    np.testing.assert_array_equal(xs0.download(), xs1.download())
    # bytes: no row-major form left, the sweeps' forms unchanged
    b0, b1 = full.layout_bytes(), lean.layout_bytes()
    assert b1["row_major"] == 0 and b0["row_major"] > 0
    assert b1["forward_form"] == b0["forward_form"] and b1["column_block_copy"] == b0["column_block_copy"]
    csr_bytes = 8 * (rows + 1) + (8 if explicit else 4) * full.nnz + 4 * rows
    held = b1["forward_form"] + b1["column_block_copy"]
    assert held <= 1.6 * csr_bytes, (held, csr_bytes)
    # whatever reads rows refuses
    for call in (lambda: lean.download(),
                 lambda: lean.ftrl_pass(g1, xs1, ctx.vec(dim), z, z, 0.1, 1.0, 1.0, 1.0),
                 lambda: lean.build_transposed(col_block=2048)):
        with pytest.raises(capi.AdmmError, match="rows"):
            call()
    # ... and the set still sweeps afterwards
    assert lean.logistic_fg(x, g1, z=z, rho=0.05) == f0
    full.destroy()
    lean.destroy()


def test_a_set_without_row_masks_keeps_its_rows(ctx):
    dim = 200                                   # fewer columns than the row masks cover: no compact form, nothing freed
    s = ctx.csr_synth(5000, dim, 12.0, seed=4).set_lean().relabel_by_frequency().build_transposed(col_block=64)
    assert s.layout_bytes()["row_major"] > 0
    rowptr, col, val, label = s.download()
    assert rowptr[-1] == s.nnz
    s.destroy()
