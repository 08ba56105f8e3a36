"""Parity of the HIP path (through the C ABI) with the CPU oracle -- runs on the MI355X box only.

Tolerances (fp64):  posterior mean <= 1e-8 relative (BASELINE.json north_star), variance <= 1e-9
absolute, Cholesky factor / v / NLML <= 1e-10 relative against the oracle's direct-distance
mode; the PyMC-faithful GEMM-distance mode is held to the 1e-8 bar.
"""
import os
from pathlib import Path

import numpy as np
import pytest

from oracle import gp_oracle as O

pytestmark = pytest.mark.gpu

GOLD = np.load(Path(__file__).resolve().parent / "golden" / "gp_goldens.npz")
CASES = sorted({k.split("/")[0] for k in GOLD.files})


def kspec(spec):
    from gumbi_amd.engine import KernelSpec

    return KernelSpec(D=spec["D"], idx_cont=spec["idx_cont"], kind=spec["kind"], ard=spec["ard"],
                      idx_lin=spec["idx_lin"], coreg=spec["coreg"], out_col=spec["out_col"], n_out=spec["n_out"],
                      hetero_noise=spec["hetero_noise"], jitter=spec["jitter"], additive=spec.get("additive", False))


def make_engine(spec, theta, X, y):
    from gumbi_amd.engine import Engine

    eng = Engine(0)
    eng.set_data(X, y)
    eng.set_kernel(kspec(spec))
    eng.set_theta(theta)
    return eng


def golden_spec(case):
    if case.startswith("composite"):
        return O.make_spec(4, [0, 1], idx_lin=[1], coreg=[(2, 3)], out_col=3, n_out=2, hetero_noise=True)
    d = GOLD[f"{case}/X"].shape[1]
    return O.make_spec(d, range(d), kind=int(GOLD[f"{case}/kind"]), ard=bool(GOLD[f"{case}/ard"]))


def rel(a, b):
    return np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(np.max(np.abs(b)), 1e-300)


# ----------------------------------------------------------------------------------------------
# block kernels
# ----------------------------------------------------------------------------------------------
def test_mfma_gemm_block_op_layouts(gpu):
    """C = beta C + alpha A B^T with an ASYMMETRIC product, so a transposed D mapping fails."""
    import torch

    from gumbi_amd.engine import Engine

    eng = Engine(0)
    rng = np.random.default_rng(0)
    m, n, k = 256, 384, 160
    A = rng.standard_normal((m, k))
    B = rng.standard_normal((n, k))
    C0 = rng.standard_normal((n, m))  # C[n + m*ldc]: row index n is the fast one
    dev = torch.device("cuda:0")
    # column-major buffers == torch tensors of the transposed shape
    tA = torch.tensor(A.T.copy(), device=dev)  # (k, m) row-major == A col-major, lda = m
    tB = torch.tensor(B.T.copy(), device=dev)
    tC = torch.tensor(C0.T.copy(), device=dev)  # (m, n) row-major == C col-major with ld n
    torch.cuda.synchronize()
    eng.blk_gemm_nt(tC.data_ptr(), n, tA.data_ptr(), m, tB.data_ptr(), n, m, n, k, -1.0, 1.0)
    torch.cuda.synchronize()
    got = tC.cpu().numpy().T  # back to (n, m)
    want = C0 - B @ A.T
    assert rel(got, want) < 1e-13
    # beta = 0 overwrite + tri skipping (tiles with tn < tm untouched)
    tC2 = torch.full((m, n), 7.0, dtype=torch.float64, device=dev)
    torch.cuda.synchronize()  # the fill runs on torch's stream, the product on the engine's own
    eng.blk_gemm_nt(tC2.data_ptr(), n, tA.data_ptr(), m, tB.data_ptr(), n, m, n, k, 2.0, 0.0, 1, 0)
    torch.cuda.synchronize()
    got2 = tC2.cpu().numpy().T
    want2 = 2.0 * B @ A.T
    # tri: every entry on or below the diagonal of the (n, m) index space is computed; block
    # tiles strictly above it are skipped (how many depends on the tile size the launch picked)
    ni, mi = np.meshgrid(np.arange(n), np.arange(m), indexing="ij")
    low = ni >= mi
    assert rel(got2[low], want2[low]) < 1e-13
    for tm in range(m // 128):
        for tn in range(n // 128):
            if tn < tm:
                assert np.all(got2[tn * 128:(tn + 1) * 128, tm * 128:(tm + 1) * 128] == 7.0)
    upper = got2[~low]
    assert np.all((upper == 7.0) | (np.abs(upper - want2[~low]) < 1e-10 * np.abs(want2).max()))


@pytest.mark.parametrize("naive", [False, True])
@pytest.mark.parametrize("nvalid", [128, 127, 112, 100, 17, 16, 8, 1])
def test_potrf_leaf_block(gpu, naive, nvalid):
    import torch

    from gumbi_amd.engine import Engine

    os.environ["GMB_LEAF_NAIVE"] = "1" if naive else "0"
    try:
        eng = Engine(0)
    finally:
        os.environ.pop("GMB_LEAF_NAIVE", None)
    rng = np.random.default_rng(nvalid)
    G = rng.standard_normal((nvalid, 200))
    S = G @ G.T / 200 + 0.5 * np.eye(nvalid)
    block = np.zeros((128, 128))
    block[:nvalid, :nvalid] = S
    P = rng.standard_normal((128 - nvalid, nvalid))  # panel rows (y row + padding rows)
    block[nvalid:, :nvalid] = P
    block[np.arange(nvalid, 128), np.arange(nvalid, 128)] = 1.0
    dev = torch.device("cuda:0")
    tA = torch.tensor(block.T.copy(), device=dev)  # column-major
    tD = torch.zeros(8 * 256, dtype=torch.float64, device=dev)
    tI = torch.zeros((128, 128), dtype=torch.float64, device=dev)
    tL = torch.zeros(1, dtype=torch.float64, device=dev)
    tinfo = torch.zeros(1, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    eng.blk_potrf(tA.data_ptr(), 128, nvalid, tD.data_ptr(), tL.data_ptr(), tinfo.data_ptr())
    eng.blk_invert(tA.data_ptr(), 128, nvalid, tD.data_ptr(), tI.data_ptr())
    torch.cuda.synchronize()
    out = tA.cpu().numpy().T
    L = np.linalg.cholesky(S)
    assert int(tinfo.cpu()[0]) == 0
    assert rel(np.tril(out[:nvalid, :nvalid]), L) < 1e-13
    if nvalid < 128:
        want_rows = np.linalg.solve(L, P.T).T  # P L^-T
        assert rel(out[nvalid:, :nvalid], want_rows) < 1e-12
    assert np.isclose(float(tL.cpu()[0]), np.sum(np.log(np.diag(L))), rtol=1e-13)
    Lpad = np.eye(128)
    Lpad[:nvalid, :nvalid] = L
    # inverses of the eight 16 x 16 diagonal sub-blocks of the identity-padded factor (column-major)
    d16 = tD.cpu().numpy().reshape(8, 16, 16).transpose(0, 2, 1)
    for s_ in range(8):
        sl = slice(16 * s_, 16 * s_ + 16)
        want = np.linalg.inv(Lpad[sl, sl])
        assert np.max(np.abs(d16[s_] - want)) < 1e-12 * np.max(np.abs(want))
        assert np.all(np.triu(d16[s_], 1) == 0.0)
    inv = tI.cpu().numpy().T
    want_inv = np.linalg.inv(Lpad)
    assert np.max(np.abs(inv - want_inv)) < 1e-11 * np.max(np.abs(want_inv))
    assert np.all(np.triu(inv, 1) == 0.0)
    # strip solve B <- B inv(L)^T on rows that are not a multiple of the workgroup's 64
    rows, ldb = 208, 256
    B = rng.standard_normal((rows, 128))
    if nvalid < 128:
        B[:, nvalid:] = 0.0
    tB = torch.zeros((128, ldb), dtype=torch.float64, device=dev)  # column-major rows x 128, ld = ldb
    tB[:, :rows] = torch.tensor(B.T.copy(), device=dev)
    guard = torch.full((128, ldb - rows), 7.0, dtype=torch.float64, device=dev)
    tB[:, rows:] = guard
    torch.cuda.synchronize()
    eng.blk_trsm(tB.data_ptr(), ldb, rows, tA.data_ptr(), 128, tD.data_ptr(), nvalid)
    torch.cuda.synchronize()
    got = tB.cpu().numpy().T
    want = np.linalg.solve(Lpad, B.T).T
    assert np.max(np.abs(got[:rows] - want)) < 1e-12 * max(1.0, np.max(np.abs(want)))
    assert np.all(got[rows:] == 7.0)


def test_potrf_leaf_flags_indefinite_block(gpu):
    import torch

    from gumbi_amd.engine import Engine

    eng = Engine(0)
    block = np.eye(128)
    block[40, 40] = -1.0
    dev = torch.device("cuda:0")
    tA = torch.tensor(block, device=dev)
    tD = torch.zeros(8 * 256, dtype=torch.float64, device=dev)
    tinfo = torch.zeros(1, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    eng.blk_potrf(tA.data_ptr(), 128, 128, tD.data_ptr(), 0, tinfo.data_ptr())
    torch.cuda.synchronize()
    assert int(tinfo.cpu()[0]) == 41


# ----------------------------------------------------------------------------------------------
# covariance build alone (gmb_blk_covariance): interior tiles take r^2 off the matrix pipe (PyMC's own
# |x|^2 + |x'|^2 - 2 x.x' expansion as one MFMA contraction), diagonal / boundary tiles the direct form
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("kind", list(O.KINDS))
@pytest.mark.parametrize("N,d", [(700, 8), (515, 1), (640, 2), (600, 3), (520, 16), (1300, 5)])
def test_covariance_build_matches_oracle_entrywise(gpu, kind, N, d):
    import torch

    X, y, ls = O.synthetic_table(N, d, seed=3 * N + d)
    spec = O.make_spec(d, range(d), kind=kind)
    theta = O.pack_theta(spec, ls, 1.3, 0.2)
    eng = make_engine(spec, theta, X, y)
    Nr, Np = (N + 1 + 127) // 128 * 128, (N + 127) // 128 * 128
    out = torch.full((Np, Nr), float("nan"), dtype=torch.float64, device="cuda:0")  # column-major Nr x Np
    torch.cuda.synchronize()
    eng.blk_covariance(out.data_ptr(), Nr)
    S = out.cpu().numpy().T  # S[i, j]
    S_dir = O.sigma_matrix(spec, theta, X, dist_mode="direct")
    S_gem = O.sigma_matrix(spec, theta, X, dist_mode="gemm")
    low = np.tril(np.ones((N, N), bool))
    eps = np.finfo(float).eps
    if kind in ("Matern12", "Exponential"):
        # kernels with a kink at r = 0 turn an r^2 error e into e / (2 r) near coinciding points (5e-10 in PyMC's own
        # expansion at r^2 = 0: sqrt(r^2 + 1e-12)); they keep the direct form on every tile
        assert np.max(np.abs(S[:N, :N][low] - S_dir[low])) < 16 * eps * 1.3**2
    else:
        # the expansion's rounding scales with |x / ls|^2; |dk/dr2| <= 3/2 for the smooth kinds
        scale2 = float(np.max(np.sum((X / ls) ** 2, axis=1)))
        tol = 40 * eps * max(1.0, scale2) * 1.3**2
        assert np.max(np.abs(S[:N, :N][low] - S_dir[low])) < tol
        assert np.max(np.abs(S[:N, :N][low] - S_gem[low])) < tol
    # the tiles that hold a diagonal entry use the direct form: agreement to the last bits of exp / sqrt there
    blk = (np.arange(N)[:, None] // 128) == (np.arange(N)[None, :] // 128)
    assert np.max(np.abs(S[:N, :N][low & blk] - S_dir[low & blk])) < 8 * np.finfo(float).eps * 1.3**2
    # y row and identity padding
    assert np.array_equal(S[N, :N], y)
    assert np.array_equal(S[N + 1:, :N], np.zeros((Nr - N - 1, N)))
    pad = S[N:, N:Np]
    assert np.array_equal(np.tril(pad), np.tril(np.eye(Nr - N, Np - N)))
    eng.close()


def test_covariance_build_with_a_nan_point_poisons_its_row_and_column(gpu):
    """Interior tiles holding a non-finite point fall back to the direct form, so NaN spreads over the
    point's whole row and column exactly as in the reference's arithmetic (and the factorisation fails)."""
    import torch

    N, d = 600, 3
    X, y, ls = O.synthetic_table(N, d, seed=11)
    X[450, 1] = np.nan
    spec = O.make_spec(d, range(d), kind="Matern52")
    eng = make_engine(spec, O.pack_theta(spec, ls, 1.0, 0.2), X, y)
    Nr, Np = (N + 1 + 127) // 128 * 128, (N + 127) // 128 * 128
    out = torch.zeros((Np, Nr), dtype=torch.float64, device="cuda:0")
    torch.cuda.synchronize()
    eng.blk_covariance(out.data_ptr(), Nr)
    S = out.cpu().numpy().T
    assert np.all(np.isnan(S[450, :451])) and np.all(np.isnan(S[450:N, 450]))
    rest = np.tril(np.ones((N, N), bool))
    rest[450, :] = rest[:, 450] = False
    assert np.all(np.isfinite(S[:N, :N][rest]))
    with pytest.raises(np.linalg.LinAlgError):
        eng.factorize()
    eng.close()


# ----------------------------------------------------------------------------------------------
# factorisation / prediction against the oracle
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("N,d,kind", [(1, 1, "ExpQuad"), (5, 2, "ExpQuad"), (127, 3, "Matern52"), (128, 4, "ExpQuad"),
                                      (129, 1, "Matern32"), (300, 8, "Matern52"), (640, 5, "Exponential"),
                                      (1000, 4, "ExpQuad"), (777, 16, "Matern12"), (513, 9, "ExpQuad")])
def test_factor_v_and_nlml_match_oracle(gpu, N, d, kind):
    X, y, ls = O.synthetic_table(N, d, seed=N + d) if N > 2 else (np.array([[0.3] * d] * N), np.array([0.7] * N), np.ones(d))
    spec = O.make_spec(d, range(d), kind=kind)
    theta = O.pack_theta(spec, ls, 1.1, 0.2)
    eng = make_engine(spec, theta, X, y)
    eng.factorize()
    L_ref, v_ref = O.factorize(spec, theta, X, y, dist_mode="direct")
    L = np.tril(eng.copy_factor())
    assert rel(L, L_ref) < 1e-10
    assert rel(eng.copy_v(), v_ref) < 1e-10
    assert np.isclose(eng.nlml(), O.nlml(spec, theta, X, y, dist_mode="direct"), rtol=1e-11, atol=1e-10)
    tm = eng.timings()
    assert tm["kbuild_ms"] > 0 and tm["chol_ms"] > 0


def test_evaluate_records_phase_events_on_small_matrices_only_when_profiling(gpu):
    """gmb_evaluate with a gradient on a matrix below 4096 rows records no per-phase events (six event records were a tenth of
    such an evaluation): kbuild_ms / chol_ms / grad_ms read 0 -- unless gmb_set_profiling is on; gmb_factorize always records."""
    X, y, ls = O.synthetic_table(300, 2, seed=5)
    spec = O.make_spec(2, range(2))
    theta = O.pack_theta(spec, ls, 1.0, 0.2)
    eng = make_engine(spec, theta, X, y)
    val, g = eng.evaluate(theta)
    tm = eng.timings()
    assert tm["kbuild_ms"] == 0 and tm["chol_ms"] == 0 and tm["grad_ms"] == 0
    eng.set_profiling(True)
    val2, g2 = eng.evaluate(theta)
    tm = eng.timings()
    assert tm["kbuild_ms"] > 0 and tm["chol_ms"] > 0 and tm["grad_ms"] > 0
    assert np.float64(val2).tobytes() == np.float64(val).tobytes() and g2.tobytes() == g.tobytes()
    eng.set_profiling(False)
    eng.factorize()
    tm = eng.timings()
    assert tm["kbuild_ms"] > 0 and tm["chol_ms"] > 0
    eng.close()


@pytest.mark.parametrize("case", CASES)
def test_predict_matches_goldens(gpu, case):
    """Committed golden vectors (oracle in PyMC-faithful mode, cross-checked with scikit-learn)."""
    spec = golden_spec(case)
    X, y, Xs, theta = (GOLD[f"{case}/{k}"] for k in ("X", "y", "Xs", "theta"))
    eng = make_engine(spec, theta, X, y)
    eng.factorize()
    mu, var = eng.predict(Xs, with_noise=True)
    assert rel(mu, GOLD[f"{case}/mu"]) < 1e-8
    assert np.max(np.abs(var - GOLD[f"{case}/var"])) < 1e-9
    mu0, var0 = eng.predict(Xs, with_noise=False)
    assert np.array_equal(mu0, mu)
    assert np.max(np.abs(var0 - GOLD[f"{case}/var_noiseless"])) < 1e-9
    assert np.isclose(eng.nlml(), float(GOLD[f"{case}/nlml"]), rtol=1e-9)
    # tighter against the oracle evaluated with the same (direct) distance formula
    mu_d, var_d = O.predict(spec, theta, X, y, Xs, with_noise=True, dist_mode="direct")
    assert rel(mu, mu_d) < 1e-10 and np.max(np.abs(var - var_d)) < 1e-11


@pytest.mark.parametrize("case", CASES)
def test_predict_gemm_form_matches_goldens(gpu, case):
    """Behind a gradient evaluation the engine holds U = L^-T of the resident factor and forms K(X*, X) L^-T as ONE product with
    a triangular operand instead of a solve (csrc/predict_form.hpp): same goldens, same tolerances, and the
    solve form on the same engine beside it."""
    spec = golden_spec(case)
    X, y, Xs, theta = (GOLD[f"{case}/{k}"] for k in ("X", "y", "Xs", "theta"))
    eng = make_engine(spec, theta, X, y)
    eng.evaluate(theta)  # objective + gradient: the factor AND its inverse stay resident
    mu, var = eng.predict(Xs, with_noise=True)
    assert eng.timings()["predict_gemm_form"] == 1
    assert rel(mu, GOLD[f"{case}/mu"]) < 1e-8 and np.max(np.abs(var - GOLD[f"{case}/var"])) < 1e-9
    mu_d, var_d = O.predict(spec, theta, X, y, Xs, with_noise=True, dist_mode="direct")
    assert rel(mu, mu_d) < 1e-10 and np.max(np.abs(var - var_d)) < 1e-11
    mu2, var2 = eng.predict(Xs, with_noise=True)  # (the transposed inverse is reused: same bits)
    assert np.array_equal(mu2, mu) and np.array_equal(var2, var)
    assert eng.set_predict_form(0) == -1
    mu_s, var_s = eng.predict(Xs, with_noise=True)
    assert eng.timings()["predict_gemm_form"] == 0
    assert rel(mu, mu_s) < 1e-11 and np.max(np.abs(var - var_s)) < 1e-12
    eng.set_predict_form(-1)
    # a factorisation without a gradient has no inverse: back to the solve, silently
    eng.factorize()
    mu_f, var_f = eng.predict(Xs, with_noise=True)
    # (another schedule factored the matrix -- gmb_factorize, not the fused launch: equal to rounding, not to the bit)
    assert eng.timings()["predict_gemm_form"] == 0 and rel(mu_f, mu_s) < 1e-12 and np.max(np.abs(var_f - var_s)) < 1e-12
    eng.close()


@pytest.mark.parametrize("N,M", [(129, 1), (1000, 127), (2321, 129), (5000, 1000)])
def test_predict_gemm_form_ragged_sizes(gpu, N, M):
    """Ragged last blocks on both sides (the inverse's padding rows and columns are zero, the cross-covariance's padding too),
    through gmb_evaluate and through gmb_factorize + gmb_nlml(grad)."""
    d = 3
    X, y, ls = O.synthetic_table(N, d, seed=N)
    spec = O.make_spec(d, range(d), kind="Matern52")
    theta = O.pack_theta(spec, ls, 1.3, 0.25)
    Xs = np.random.default_rng(M).standard_normal((M, d))
    mu_r, var_r = O.predict(spec, theta, X, y, Xs, dist_mode="direct")
    eng = make_engine(spec, theta, X, y)
    eng.evaluate(theta)
    mu, var = eng.predict(Xs)
    assert eng.timings()["predict_gemm_form"] == 1
    assert rel(mu, mu_r) < 1e-10 and np.max(np.abs(var - var_r)) < 1e-11
    eng.factorize()
    eng.nlml(grad=True)
    by_tiles = -(-N // 128) >= 6  # (below six block columns the separate gradient call takes the launch tree: no U by rows)
    mu2, var2 = eng.predict(Xs)
    assert eng.timings()["predict_gemm_form"] == (1 if by_tiles else 0)
    assert rel(mu2, mu_r) < 1e-10 and np.max(np.abs(var2 - var_r)) < 1e-11
    # new hyper-parameters: nothing of the old inverse may be used
    theta2 = O.pack_theta(spec, ls * 1.5, 0.9, 0.3)
    eng.set_theta(theta2)
    with pytest.raises(ValueError):
        eng.predict(Xs)
    eng.evaluate(theta2)
    mu3, var3 = eng.predict(Xs)
    mu_r3, var_r3 = O.predict(spec, theta2, X, y, Xs, dist_mode="direct")
    assert rel(mu3, mu_r3) < 1e-10 and np.max(np.abs(var3 - var_r3)) < 1e-11
    eng.close()


@pytest.mark.parametrize("nv", [1, 15, 16, 17, 47, 100, 127, 128])
def test_ragged_last_block_extents(gpu, nv):
    """The persistent launches contract only the real part of the ragged last block (sixteen-row groups of
    the last block ROW -- the y row included -- and sixteen-column groups of the last block COLUMN of the inverse and of the
    predict solve): every extent class against the oracle, through gmb_factorize + gmb_predict (tile Cholesky, tile solve),
    gmb_evaluate (fused launch) and the GEMM-form prediction behind it; nv = 128: the y row alone is the last block row."""
    N, d, M = 128 * 7 + nv, 3, 4200  # (33 row tiles of test points: the tile triangular solve)
    X, y, ls = O.synthetic_table(N, d, seed=100 + nv)
    spec = O.make_spec(d, range(d), kind="Matern32")
    theta = O.pack_theta(spec, ls, 1.2, 0.3)
    Xs = np.random.default_rng(nv).standard_normal((M, d))
    mu_r, var_r = O.predict(spec, theta, X, y, Xs, dist_mode="direct")
    val_r, grad_r = O.nlml_and_grad(spec, theta, X, y)
    eng = make_engine(spec, theta, X, y)
    eng.factorize()
    assert abs(eng.nlml() - val_r) < 1e-10 * abs(val_r)
    mu, var = eng.predict(Xs)
    assert eng.timings()["predict_gemm_form"] == 0
    assert rel(mu, mu_r) < 1e-10 and np.max(np.abs(var - var_r)) < 1e-11
    L = eng.copy_factor(0, N, 0, N)
    S = O.sigma_matrix(spec, theta, X)
    assert np.max(np.abs(np.tril(L) @ np.tril(L).T - S)) < 1e-11 * np.max(np.abs(S))
    val, grad = eng.evaluate(theta)
    assert abs(val - val_r) < 1e-10 * abs(val_r) and rel(grad, grad_r) < 1e-8
    assert rel(eng.copy_alpha(), np.linalg.solve(S, y)) < 1e-8
    mu2, var2 = eng.predict(Xs)
    assert eng.timings()["predict_gemm_form"] == 1
    assert rel(mu2, mu_r) < 1e-10 and np.max(np.abs(var2 - var_r)) < 1e-11
    eng.close()


def test_reserve_allocates_ahead_and_changes_no_result(gpu):
    """gmb_reserve: the factor / Sigma^-1 / prediction buffers allocated before their first use -- resident bytes go up by what
    they need, the following calls allocate nothing more of them and return what they return without it."""
    N, d, M = 1500, 3, 700
    X, y, ls = O.synthetic_table(N, d, seed=77)
    spec = O.make_spec(d, range(d))
    theta = O.pack_theta(spec, ls, 1.0, 0.2)
    Xs = np.random.default_rng(1).standard_normal((M, d))
    ref = make_engine(spec, theta, X, y)
    val_r, grad_r = ref.evaluate(theta)
    mu_r, var_r = ref.predict(Xs)
    ref.close()
    eng = make_engine(spec, theta, X, y)
    before = eng.resident_bytes()
    eng.reserve(gradient=True, M=M)
    after = eng.resident_bytes()
    Np, Nr = 1536, 1536
    assert after - before >= 8 * (Nr * Np + Np * Np + 768 * Np)
    val, grad = eng.evaluate(theta)
    mu, var = eng.predict(Xs)
    assert val == val_r and np.array_equal(grad, grad_r) and np.array_equal(mu, mu_r) and np.array_equal(var, var_r)
    fresh = make_engine(spec, theta, X, y)
    with pytest.raises(ValueError):
        fresh.reserve(M=-1)
    fresh.close()
    eng.close()


def test_predict_edge_cases(gpu):
    X, y, ls = O.synthetic_table(200, 3, seed=5)
    spec = O.make_spec(3, range(3))
    theta = O.pack_theta(spec, ls, 1.0, 0.2)
    eng = make_engine(spec, theta, X, y)
    with pytest.raises(ValueError):
        eng.predict(X[:3])  # not factorised yet
    eng.factorize()
    mu, var = eng.predict(np.zeros((0, 3)))
    assert mu.shape == (0,) and var.shape == (0,)
    with pytest.raises(ValueError):
        eng.predict(np.zeros((4, 2)))
    for M in (1, 127, 128, 129, 1000):  # ragged tile edges
        Xs = np.random.default_rng(M).standard_normal((M, 3))
        mu, var = eng.predict(Xs)
        mu_r, var_r = O.predict(spec, theta, X, y, Xs, dist_mode="direct")
        assert rel(mu, mu_r) < 1e-10 and np.max(np.abs(var - var_r)) < 1e-11
    # at the training inputs the noiseless posterior interpolates towards y
    mu, var = eng.predict(X, with_noise=False)
    assert np.all(var > -1e-12) and np.all(var < 0.2**2 + 1e-6)
    # refit with other hyper-parameters reuses the buffers
    theta2 = O.pack_theta(spec, ls * 1.7, 0.8, 0.35)
    eng.set_theta(theta2)
    with pytest.raises(ValueError):
        eng.predict(X[:3])
    eng.factorize()
    mu, _ = eng.predict(X[:50])
    assert rel(mu, O.predict(spec, theta2, X, y, X[:50], dist_mode="direct")[0]) < 1e-10


def test_predict_with_a_nan_test_point_poisons_only_that_point(gpu):
    """Cross-covariance tiles holding a non-finite test point leave the matrix-pipe form for the direct loop:
    that point's mean / variance come out NaN (as the reference's arithmetic gives), every other point is untouched."""
    N, d, M = 700, 4, 400
    X, y, ls = O.synthetic_table(N, d, seed=23)
    spec = O.make_spec(d, range(d), kind="Matern32")
    eng = make_engine(spec, O.pack_theta(spec, ls, 1.0, 0.2), X, y)
    eng.factorize()
    Xs = np.random.default_rng(3).standard_normal((M, d))
    mu0, var0 = eng.predict(Xs)
    Xs_bad = Xs.copy()
    Xs_bad[200, 2] = np.nan
    mu, var = eng.predict(Xs_bad)
    assert np.isnan(mu[200]) and np.isnan(var[200])
    keep = np.arange(M) != 200
    # (its 127 tile-row neighbours take the direct loop too: same values to the rounding of r^2's two forms)
    assert np.allclose(mu[keep], mu0[keep], rtol=0, atol=1e-12) and np.allclose(var[keep], var0[keep], rtol=0, atol=1e-12)
    eng.close()


def test_colliding_inputs_and_tiny_noise(gpu):
    """Exactly repeated rows make K rank-deficient: only sigma^2 + jitter keeps Sigma positive definite.
    The factor, NLML, gradient and predictions must still follow the oracle (the jitter is part of
    the model, pymc/GP.py:580 -> 1e-6 on the diagonal), with the tolerance scaled by the conditioning."""
    X, y, ls = O.synthetic_table(260, 2, seed=13)
    X[130:] = X[:130]                      # every point appears twice
    y[130:] = y[:130] + 0.01
    spec = O.make_spec(2, range(2))
    theta = O.pack_theta(spec, ls, 1.0, 1e-2)   # sigma^2 = 1e-4: cond(Sigma) ~ 2.6e6
    eng = make_engine(spec, theta, X, y)
    eng.factorize()
    val, g = eng.nlml(grad=True)
    val_r, g_r = O.nlml_and_grad(spec, theta, X, y, dist_mode="direct")
    assert np.isclose(val, val_r, rtol=1e-9)
    assert np.max(np.abs(g - g_r)) < 1e-6 * max(1.0, np.max(np.abs(g_r)))
    eng.factorize()
    Xs = np.random.default_rng(0).standard_normal((300, 2))
    mu, var = eng.predict(Xs)
    mu_r, var_r = O.predict(spec, theta, X, y, Xs, dist_mode="direct")
    assert rel(mu, mu_r) < 1e-8 and np.max(np.abs(var - var_r)) < 1e-8
    # sigma -> 0: jitter alone (1e-6) on a doubled data set -- still factorises, like the reference does
    eng.set_theta(O.pack_theta(spec, ls, 1.0, 1e-9))
    eng.factorize()
    assert np.isfinite(eng.nlml())


def test_not_positive_definite_raises_linalgerror(gpu):
    X, y, ls = O.synthetic_table(150, 2, seed=9)
    X[77, 0] = np.nan
    spec = O.make_spec(2, range(2))
    eng = make_engine(spec, O.pack_theta(spec, ls, 1.0, 0.2), X, y)
    with pytest.raises(np.linalg.LinAlgError):
        eng.factorize()
    assert eng.notpd_index() >= 0
    with pytest.raises(ValueError):
        eng.set_theta(O.pack_theta(spec, [-1.0, 1.0], 1.0, 0.2))
    with pytest.raises(ValueError):
        eng.set_theta([1.0, 1.0])


@pytest.mark.parametrize("kind", list(O.KINDS))
@pytest.mark.parametrize("ard", [True, False])
def test_nlml_gradient_matches_oracle(gpu, kind, ard):
    N, d = 333, 3
    X, y, ls = O.synthetic_table(N, d, seed=21)
    spec = O.make_spec(d, range(d), kind=kind, ard=ard)
    theta = O.pack_theta(spec, ls if ard else [1.2], 1.2, 0.25)
    eng = make_engine(spec, theta, X, y)
    eng.factorize()
    Xs = np.random.default_rng(4).standard_normal((40, d))
    mu0, var0 = eng.predict(Xs)
    L0, v0 = np.tril(eng.copy_factor()), eng.copy_v()
    val, g = eng.nlml(grad=True)
    val_r, g_r = O.nlml_and_grad(spec, theta, X, y, dist_mode="direct")
    assert np.isclose(val, val_r, rtol=1e-11)
    assert np.max(np.abs(g - g_r)) < 1e-8 * max(1.0, np.max(np.abs(g_r)))
    # the gradient borrows the factor buffer's upper triangle and diagonal tiles for U = L^-T and puts the
    # diagonal blocks back: the factorisation is intact afterwards -- same factor, same predictions, bit for bit,
    # and a second gradient evaluation gives the same numbers without a re-factorisation
    assert eng.factor_is_current()
    assert np.array_equal(np.tril(eng.copy_factor()), L0) and np.array_equal(eng.copy_v(), v0)
    mu1, var1 = eng.predict(Xs)
    assert np.array_equal(mu1, mu0) and np.array_equal(var1, var0)
    val2, g2 = eng.nlml(grad=True)
    assert val2 == val and np.max(np.abs(g2 - g)) < 1e-12 * max(1.0, np.max(np.abs(g)))
    eng.set_theta(theta)
    assert not eng.factor_is_current()


@pytest.mark.parametrize("kind", ["ExpQuad", "Matern52", "Matern32"])
@pytest.mark.parametrize("d,ard", [(1, True), (2, True), (3, False), (5, True), (8, True), (11, True), (16, False)])
def test_interior_tiles_of_the_trace_reductions_match_the_oracle(gpu, kind, d, ard):
    """grad_interior_kernel (csrc/gradient.hpp): on full tiles below the diagonal the per-dimension sums run as r^2 and
    G . [X | X^2] contractions on the matrix pipe.  Every compile-time dimension count (1, 2: norms in the contraction's spare
    slots; 4, 8, 16: in the accumulator; 16: two feature accumulators), ARD and shared lengthscales, length scales short enough
    that the expansion's cancellation would show -- gradient and NLML against the oracle's direct-difference form.
    (N = 2200: 171 tiles -- up to 128 tiles every tile goes through the direct loops, four workgroups per tile.)"""
    N = 2200
    X, y, ls = O.synthetic_table(N, d, seed=40 + d)
    spec = O.make_spec(d, range(d), kind=kind, ard=ard)
    theta = O.pack_theta(spec, 0.6 * ls if ard else [0.8], 1.3, 0.2)
    eng = make_engine(spec, theta, X, y)
    val, g = eng.evaluate(theta)
    val_r, g_r = O.nlml_and_grad(spec, theta, X, y, dist_mode="direct")
    assert np.isclose(val, val_r, rtol=1e-11) and np.max(np.abs(g - g_r)) < 1e-8 * max(1.0, np.max(np.abs(g_r)))
    val2, g2 = eng.evaluate(theta)
    assert g2.tobytes() == g.tobytes()
    eng.close()


@pytest.mark.parametrize("two_outputs,lin,hetero,n", [(True, True, True, 90), (False, True, False, 300), (True, False, False, 200)])
def test_additive_model_matches_oracle(gpu, two_outputs, lin, hetero, n):
    """specify_model(additive=True) (pymc/GP.py:732-754): a global kernel plus one kernel per categorical
    dim, each times that dim's coregion table (and the output table).  The engine builds and
    differentiates the sum term by term (accumulating passes of the same kernels)."""
    from test_oracle import additive_problem

    spec, theta, X, y = additive_problem(n=n, two_outputs=two_outputs, lin=lin, hetero=hetero)
    eng = make_engine(spec, theta, X, y)
    eng.factorize()
    L_ref, v_ref = O.factorize(spec, theta, X, y, dist_mode="direct")
    assert rel(np.tril(eng.copy_factor()), L_ref) < 1e-10
    assert rel(eng.copy_v(), v_ref) < 1e-9
    Xs = X[::3].copy()
    Xs[:, 0] += 0.17
    for with_noise in (True, False):
        mu, var = eng.predict(Xs, with_noise=with_noise)
        mu_r, var_r = O.predict(spec, theta, X, y, Xs, with_noise=with_noise, dist_mode="direct")
        assert rel(mu, mu_r) < 1e-8 and np.max(np.abs(var - var_r)) < 1e-9
    val, g = eng.nlml(grad=True)
    val_r, g_r = O.nlml_and_grad(spec, theta, X, y, dist_mode="direct")
    assert np.isclose(val, val_r, rtol=1e-10)
    assert np.max(np.abs(g - g_r)) < 1e-8 * max(1.0, np.max(np.abs(g_r)))


# ----------------------------------------------------------------------------------------------
# device tests of the composite gradient and of the pairwise-distance reduction (restored: ADVICE r02)
# ----------------------------------------------------------------------------------------------
def test_composite_model_gradient_and_prediction(gpu):
    """Linear x coregion x two outputs x heteroskedastic noise on ONE GPU: every accumulator class of the fused
    trace reductions (lengthscales, eta, tau, c, coregion tables, sigma, noise table) against the golden gradient."""
    case = "composite_N140"
    spec = golden_spec(case)
    X, y, Xs, theta = (GOLD[f"{case}/{k}"] for k in ("X", "y", "Xs", "theta"))
    eng = make_engine(spec, theta, X, y)
    eng.factorize()
    val, g = eng.nlml(grad=True)
    assert np.isclose(val, float(GOLD[f"{case}/nlml"]), rtol=1e-10)
    g_r = GOLD[f"{case}/grad"]
    assert np.max(np.abs(g - g_r)) < 1e-8 * max(1.0, np.max(np.abs(g_r)))
    mu, var = eng.predict(Xs, with_noise=True)  # the factor survives the gradient
    mu_r, var_r = O.predict(spec, theta, X, y, Xs, with_noise=True, dist_mode="direct")
    assert rel(mu, mu_r) < 1e-8 and np.max(np.abs(var - var_r)) < 1e-9
    eng.close()


def test_ls_limits_joint_matches_oracle(gpu):
    """``gmb_ls_limits`` (ls_limits_kernel): the joint d-dimensional pairwise extrema of ``parse_ls_limits(ARD=False)``
    (gp_utils.py:34-46) and the per-column ones, against scipy's pdist; N not a multiple of the tile, a coinciding pair."""
    from scipy.spatial.distance import pdist

    from gumbi_amd.engine import ls_limits
    from gumbi_amd.utils.gp_utils import parse_ls_limits

    rng = np.random.default_rng(4)
    for n, d in ((700, 4), (1031, 3), (129, 8)):
        X = rng.standard_normal((n, d))
        X[10] = X[3]
        lo, hi = ls_limits(X, ard=False)
        dd = pdist(X)
        assert np.isclose(lo[0], dd[dd != 0].min(), rtol=1e-13) and np.isclose(hi[0], dd.max(), rtol=1e-13)
        lo, hi = ls_limits(X, ard=True)  # raw extrema (the 0.01 floor is applied by parse_ls_limits)
        for j in range(d):
            dj = pdist(X[:, [j]])
            assert np.isclose(lo[j], dj[dj != 0].min(), rtol=1e-13) and np.isclose(hi[j], dj.max(), rtol=1e-13)
        lo2, hi2 = parse_ls_limits(X, ARD=False)
        assert np.isclose(lo2[0], max(dd[dd != 0].min(), 0.01)) and np.isclose(hi2[0], dd.max())
    lo, hi = ls_limits(np.ones((5, 2)), ard=False)
    assert lo[0] == -1.0  # every pair coincides


# ----------------------------------------------------------------------------------------------
# bit-reproducibility (deterministic reductions)
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("model", ["matern_ard", "expquad_shared", "composite"])
def test_evaluations_are_bit_reproducible(gpu, model):
    """No floating-point atomics whose order depends on scheduling: |v|^2, the log-determinant and every partial
    of the fused trace reductions are summed in a fixed order (two-stage sums), so two evaluations at one theta --
    on one engine and on a fresh one -- give the SAME BITS for the NLML, its gradient and the predictions."""
    if model == "composite":
        case = "composite_N140"
        spec = golden_spec(case)
        X, y, Xs, theta = (GOLD[f"{case}/{k}"] for k in ("X", "y", "Xs", "theta"))
        reps = 9
        rng = np.random.default_rng(1)
        X = np.concatenate([X] * reps)
        X[:, spec["idx_cont"]] += 0.05 * rng.standard_normal((len(X), len(spec["idx_cont"])))
        y = np.concatenate([y] * reps) + 0.1 * rng.standard_normal(len(X))
    else:
        N, d = 3000, 6
        X, y, ls = O.synthetic_table(N, d, seed=3)
        ard = model == "matern_ard"
        spec = O.make_spec(d, range(d), kind="Matern52" if ard else "ExpQuad", ard=ard)
        theta = O.pack_theta(spec, ls if ard else [1.3], 1.2, 0.25)
        Xs = np.random.default_rng(2).standard_normal((300, d))
    runs = []
    for fresh in range(2):
        eng = make_engine(spec, theta, X, y)
        for _ in range(2):
            eng.factorize()
            val, g = eng.nlml(grad=True)
            mu, var = eng.predict(Xs)
            runs.append((np.float64(val).tobytes(), g.tobytes(), mu.tobytes(), var.tobytes()))
        eng.close()
    assert all(r == runs[0] for r in runs[1:])


def test_map_fits_are_bit_reproducible(gpu):
    """Two ``find_MAP`` runs of the same model trace the same objective values bit for bit and end at the same theta."""
    import pandas as pd

    import gumbi_amd as gmb

    X, y, _ = O.synthetic_table(2500, 3, seed=6)
    df = pd.DataFrame(X, columns=["a", "b", "c"])
    df["y"] = y
    traces = []
    for _ in range(2):
        gp = gmb.GP(gmb.DataSet(df, outputs=["y"]), outputs=["y"])
        gp.fit(continuous_dims=["a", "b", "c"], continuous_kernel="Matern32")
        traces.append((np.array(gp.nlml_trace).tobytes(), gp._theta_fitted.tobytes(), gp.n_eval))
        gp.engine.close()
    assert traces[0] == traces[1] and traces[0][2] > 5


@pytest.mark.parametrize("N", [2100, 5000])
def test_cholesky_schedules_agree(gpu, N):
    """``GMB_CHOL_SCHEME`` -- one of the three environment switches the product library keeps (plain recursion
    on one stream / masked look-ahead with the panel chain beside the CU-masked trailing updates): the two
    schedules order the same trailing updates differently, so factor, v and NLML agree to rounding, and both
    with the oracle."""
    d = 5
    X, y, ls = O.synthetic_table(N, d, seed=12)
    spec = O.make_spec(d, range(d), kind="Matern32")
    theta = O.pack_theta(spec, ls, 0.9, 0.2)
    got = {}
    try:
        for scheme in ("0", "2", "3", "4"):  # (4: the recursion with its bottom panels on the tile kernel)
            os.environ["GMB_CHOL_SCHEME"] = scheme
            eng = make_engine(spec, theta, X, y)
            eng.factorize()
            got[scheme] = (np.tril(eng.copy_factor()), eng.copy_v(), eng.nlml())
            eng.close()
    finally:
        os.environ.pop("GMB_CHOL_SCHEME", None)
    L_ref, v_ref = O.factorize(spec, theta, X, y, dist_mode="direct")
    for scheme in got:
        assert rel(got[scheme][0], L_ref) < 1e-10 and rel(got[scheme][1], v_ref) < 1e-10
    assert rel(got["0"][0], got["2"][0]) < 1e-12 and abs(got["0"][2] - got["2"][2]) < 1e-9 * abs(got["2"][2])
    assert rel(got["0"][0], got["3"][0]) < 1e-12 and abs(got["0"][2] - got["3"][2]) < 1e-9 * abs(got["3"][2])
    assert rel(got["0"][0], got["4"][0]) < 1e-12 and abs(got["0"][2] - got["4"][2]) < 1e-9 * abs(got["4"][2])


@pytest.mark.parametrize("N", [300, 1153, 2304, 2321, 4100])
def test_recursion_with_tile_panels_matches_oracle(gpu, N):
    """The large matrices' schedule (plain recursion; panels of <= 16 block columns with every row below them as ONE launch of
    the tile kernel each, csrc/engine.hip: chol_tiles_panel -- VERDICT r05 item 7), forced at sizes the oracle factors in
    seconds: factor, v, NLML, gradient and predictions against the oracle; ragged sizes, N % 128 == 0 (the y row alone is the
    last block row), fewer block columns than one panel; twice = the same bits (the failing row of a matrix that is not
    positive definite: test_tile_cholesky_reports_the_first_bad_pivot_like_the_recursion)."""
    d = 3
    X, y, ls = O.synthetic_table(N, d, seed=31)
    spec = O.make_spec(d, range(d), kind="Matern52")
    theta = O.pack_theta(spec, ls, 1.1, 0.25)
    eng = make_engine(spec, theta, X, y)
    assert eng.set_chol_scheme(eng.CHOL_RECURSION_TILE_PANELS) == eng.CHOL_BY_SIZE
    eng.set_profiling(True)
    eng.factorize()
    tm = eng.timings()
    assert tm["total_chol_panel_tile_launches"] >= 1 and tm["total_chol_tile_launches"] == 0
    L_ref, v_ref = O.factorize(spec, theta, X, y, dist_mode="direct")
    L = np.tril(eng.copy_factor())
    assert rel(L, L_ref) < 1e-10 and rel(eng.copy_v(), v_ref) < 1e-10
    val_r, grad_r = O.nlml_and_grad(spec, theta, X, y)
    val, grad = eng.nlml(grad=True)
    assert abs(val - val_r) < 1e-10 * abs(val_r) and rel(grad, grad_r) < 1e-8
    Xs = np.random.default_rng(N).standard_normal((300, d))
    mu, var = eng.predict(Xs)
    mu_r, var_r = O.predict(spec, theta, X, y, Xs, dist_mode="direct")
    assert rel(mu, mu_r) < 1e-10 and np.max(np.abs(var - var_r)) < 1e-11
    eng.factorize()
    assert np.array_equal(np.tril(eng.copy_factor()), L)  # run to run: the same bits
    eng.close()


# ----------------------------------------------------------------------------------------------
# persistent tile Cholesky (csrc/chol_tiles.hpp): the whole factorisation in ONE launch
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("N", [2, 5, 127, 128, 129, 640, 1000, 2560, 5200, 6016])
def test_tile_cholesky_matches_oracle(gpu, N):
    """Factor, v, log-det (NLML), gradient and predictions of the tile schedule against the oracle: ragged sizes, sizes with
    and without a separate y block row (N % 128 == 0), one to 47 block columns (5200 and 6016 are inside the range the
    engine picks the schedule for by itself; the others are forced)."""
    d = 3
    X, y, ls = O.synthetic_table(N, d, seed=21)
    spec = O.make_spec(d, range(d), kind="Matern52")
    theta = O.pack_theta(spec, ls, 1.1, 0.25)
    eng = make_engine(spec, theta, X, y)
    if N < 5000:
        eng.set_chol_scheme(eng.CHOL_TILES)
    eng.chol_task_trace(1)
    eng.factorize()
    tr = eng.chol_task_trace(0)
    assert tr is not None, "the factorisation did not run on the tile kernel"
    tiles, stamps = tr
    nct, nrt = (N + 127) // 128, (N + 1 + 127) // 128
    assert len(tiles) == nct * nrt - nct * (nct - 1) // 2 and (np.diff(stamps, axis=1) >= 0).all()
    L_ref, v_ref = O.factorize(spec, theta, X, y, dist_mode="direct")
    nr = min(N, 1500)
    L = eng.copy_factor(r0=N - nr, nr=nr)
    assert rel(np.tril(L, N - nr), np.tril(L_ref[N - nr:], N - nr)) < 1e-10 and rel(eng.copy_v(), v_ref) < 1e-10
    val_r, grad_r = O.nlml_and_grad(spec, theta, X, y)
    val, grad = eng.nlml(grad=True)
    assert abs(val - val_r) < 1e-10 * max(1.0, abs(val_r)) and rel(grad, grad_r) < 1e-8
    # 200 test points: the recursion's triangular solve; 1500: the persistent tile solve (trsm_tiles_kernel, 8 .. 96 row tiles)
    for M in (200, 1500):
        Xs = np.random.default_rng(3).standard_normal((M, d))
        mu, var = eng.predict(Xs)
        mu_r, var_r = O.predict(spec, theta, X, y, Xs, dist_mode="direct")
        assert rel(mu, mu_r) < 1e-8 and np.max(np.abs(var - var_r)) < 1e-9
        mu2, var2 = eng.predict(Xs)
        assert mu2.tobytes() == mu.tobytes() and var2.tobytes() == var.tobytes()  # bit-reproducible
    eng.close()


def test_tile_cholesky_is_bit_reproducible(gpu):
    """Every tile is written once by its owner and every contraction runs in k order in one accumulator: whatever order
    the workgroups draw their tickets in, the factor comes out with the same bits -- across repeats and fresh engines."""
    N, d = 5400, 4
    X, y, ls = O.synthetic_table(N, d, seed=8)
    spec = O.make_spec(d, range(d), kind="ExpQuad")
    theta = O.pack_theta(spec, ls, 1.0, 0.2)
    runs = []
    for _ in range(2):
        eng = make_engine(spec, theta, X, y)
        eng.set_chol_scheme(eng.CHOL_TILES)
        for _ in range(3):
            eng.factorize()
            runs.append((np.tril(eng.copy_factor(r0=N - 700, nr=700), N - 700).tobytes(), eng.copy_v().tobytes(), np.float64(eng.nlml()).tobytes()))
        eng.close()
    assert all(r == runs[0] for r in runs[1:])


def test_tile_cholesky_reports_the_first_bad_pivot_like_the_recursion(gpu):
    """A NaN point and an indefinite matrix: LinAlgError with the same global row as the stream schedules report, and the
    engine factorises a well-posed problem afterwards (no flag or ticket state survives a failed launch)."""
    N = 1500
    X, y, ls = O.synthetic_table(N, 2, seed=9)
    spec = O.make_spec(2, range(2))
    theta = O.pack_theta(spec, ls, 1.0, 0.2)
    Xbad = X.copy()
    Xbad[777, 0] = np.nan
    rows = {}
    for scheme in (0, 3, 4):  # (4: the recursion's bottom panels on the tile kernel)
        eng = make_engine(spec, theta, Xbad, y)
        eng.set_chol_scheme(scheme)
        with pytest.raises(np.linalg.LinAlgError):
            eng.factorize()
        rows[scheme] = eng.notpd_index()
        eng.set_data(X, y)
        eng.set_theta(theta)
        eng.factorize()
        assert np.isfinite(eng.nlml())
        eng.close()
    assert rows[0] == rows[3] == rows[4] >= 0


def test_tile_cholesky_gives_up_on_a_lost_tile_instead_of_hanging(gpu):
    """Fault injection: the first diagonal tile is never computed, so every other task of the launch ends up waiting for it.
    The waits are bounded -- the launch drains after its time-out and gmb_factorize fails loudly (GumbiHipError, status
    GMB_EHIP) within seconds; the engine then factorises the same problem correctly (fresh flags and tickets every launch)."""
    import time

    from gumbi_amd.engine import GumbiHipError

    N, d = 3000, 3
    X, y, ls = O.synthetic_table(N, d, seed=4)
    spec = O.make_spec(d, range(d))
    theta = O.pack_theta(spec, ls, 1.0, 0.2)
    eng = make_engine(spec, theta, X, y)
    eng.set_chol_scheme(eng.CHOL_TILES)
    eng.factorize()
    ref = eng.nlml()
    eng.debug_lose_tickets(1)
    t0 = time.time()
    with pytest.raises(GumbiHipError, match="time-out"):
        eng.factorize()
    assert 1.0 < time.time() - t0 < 30.0
    eng.factorize()
    assert eng.nlml() == ref
    eng.close()


# ----------------------------------------------------------------------------------------------
# persistent evaluation launch (csrc/eval_tiles.hpp): Cholesky + L^-T + Sigma^-1 as tile tasks of ONE launch
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("N,kind", [(130, "ExpQuad"), (1000, "Matern52"), (1024, "ExpQuad"), (2689, "Matern32"), (5200, "ExpQuad")])
def test_evaluation_schedules_agree_with_each_other_and_the_oracle(gpu, N, kind):
    """The gradient's inverse and Sigma^-1 as a tree of GEMM launches (scheme 0), as tile tasks behind the factorisation (1) and
    fused into the factorisation's launch (2): NLML, gradient and alpha against the oracle and against each other; ragged
    sizes and a size with a separate y block row (N % 128 == 0); the factor serves a prediction afterwards (L is left intact);
    the tile schedules give the same bits whether the gradient rode with the factorisation or followed it."""
    d = 3
    X, y, ls = O.synthetic_table(N, d, seed=31)
    spec = O.make_spec(d, range(d), kind=kind)
    theta = O.pack_theta(spec, ls, 1.15, 0.22)
    Xs = np.random.default_rng(2).standard_normal((150, d))
    val_r, grad_r = O.nlml_and_grad(spec, theta, X, y)
    mu_r, var_r = O.predict(spec, theta, X, y, Xs, dist_mode="direct")
    L_ref, v_ref = O.factorize(spec, theta, X, y, dist_mode="direct")
    import scipy.linalg

    alpha_r = scipy.linalg.solve_triangular(L_ref, v_ref, lower=True, trans="T")
    eng = make_engine(spec, theta, X, y)
    eng.set_chol_scheme(eng.CHOL_TILES)
    got = {}
    for scheme in (eng.GRAD_LAUNCH_TREE, eng.GRAD_TILES, eng.GRAD_FUSED):
        eng.set_grad_scheme(scheme)
        val, g = eng.evaluate(theta)
        alpha = eng.copy_alpha()
        assert eng.factor_is_current()
        mu, var = eng.predict(Xs)
        assert abs(val - val_r) < 1e-10 * max(1.0, abs(val_r)) and rel(g, grad_r) < 1e-8 and rel(alpha, alpha_r) < 1e-8
        assert rel(mu, mu_r) < 1e-8 and np.max(np.abs(var - var_r)) < 1e-9
        got[scheme] = (val, g, alpha, mu, var)
        if scheme != eng.GRAD_LAUNCH_TREE:  # the launch behind a factor that is final: same task bodies, same bits
            eng.factorize()
            val2, g2 = eng.nlml(grad=True)
            assert np.float64(val2).tobytes() == np.float64(val).tobytes() and g2.tobytes() == g.tobytes()
            assert eng.copy_alpha().tobytes() == alpha.tobytes()
    a, b, c = (got[s] for s in (eng.GRAD_LAUNCH_TREE, eng.GRAD_TILES, eng.GRAD_FUSED))
    assert rel(a[1], b[1]) < 1e-10 and b[1].tobytes() == c[1].tobytes() and b[2].tobytes() == c[2].tobytes()
    assert b[3].tobytes() == c[3].tobytes() and b[4].tobytes() == c[4].tobytes()
    eng.close()


@pytest.mark.parametrize("N,kind", [(300, "ExpQuad"), (2689, "Matern52"), (5200, "ExpQuad"), (9300, "Matern32")])
def test_paired_sigma_inverse_tasks_change_no_bit(gpu, N, kind):
    """gmb_set_eval_pairs: Sigma^-1 as 128 x 256 tasks (two tiles per pass over the task's own block row of U; by default for
    matrices of >= 72 block columns) against the 128 x 128 form -- every output element is the same sum in the same k order in
    one accumulator, so NLML, gradient and alpha carry the same bits; in the fused launch and behind a final factor; ragged sizes
    (the cut last k-block), pairs that end on a diagonal tile (I = J + 1)."""
    d = 3
    X, y, ls = O.synthetic_table(N, d, seed=43)
    spec = O.make_spec(d, range(d), kind=kind)
    theta = O.pack_theta(spec, ls, 1.05, 0.3)
    eng = make_engine(spec, theta, X, y)
    got = {}
    for mode in (0, 1, 0, 1):
        assert eng.set_eval_pairs(mode) in (-1, 0, 1)
        val, g = eng.evaluate(theta)
        alpha = eng.copy_alpha()
        eng.factorize()
        val2, g2 = eng.nlml(grad=True)
        rec = (np.float64(val).tobytes(), g.tobytes(), alpha.tobytes(), np.float64(val2).tobytes(), g2.tobytes())
        assert got.setdefault(mode, rec) == rec
    assert got[0] == got[1]
    if N <= 5200:
        val_r, grad_r = O.nlml_and_grad(spec, theta, X, y)
        assert abs(np.frombuffer(got[1][0])[0] - val_r) < 1e-10 * abs(val_r) and rel(np.frombuffer(got[1][1]), grad_r) < 1e-8
    eng.close()


def test_fused_evaluation_is_bit_reproducible_and_survives_a_lost_tile(gpu):
    """Whatever order the workgroups draw the tickets of the fused launch in, NLML / gradient / alpha come out with the same
    bits (every tile written once, contractions in k order); with the first diagonal tile withheld the launch gives up after
    its time-out (GumbiHipError), and the next evaluation is bit-identical to the ones before."""
    import time

    from gumbi_amd.engine import GumbiHipError

    N, d = 3300, 4
    X, y, ls = O.synthetic_table(N, d, seed=12)
    spec = O.make_spec(d, range(d), kind="Matern52")
    theta = O.pack_theta(spec, ls, 0.9, 0.3)
    runs = []
    for _ in range(2):
        eng = make_engine(spec, theta, X, y)
        for _ in range(3):
            val, g = eng.evaluate(theta)
            runs.append((np.float64(val).tobytes(), g.tobytes(), eng.copy_alpha().tobytes()))
        eng.close()
    assert all(r == runs[0] for r in runs[1:])
    eng = make_engine(spec, theta, X, y)
    eng.debug_lose_tickets(1)
    t0 = time.time()
    with pytest.raises(GumbiHipError, match="time-out"):
        eng.evaluate(theta)
    assert 1.0 < time.time() - t0 < 30.0
    val, g = eng.evaluate(theta)
    assert (np.float64(val).tobytes(), g.tobytes(), eng.copy_alpha().tobytes()) == runs[0]
    eng.close()


def test_composite_model_gradient_through_the_fused_launch(gpu):
    """Linear x coregion x two outputs x heteroskedastic noise at a size the fused launch takes by itself: every partial of the
    gradient (tables included) against the oracle -- the reductions read Sigma^-1 and alpha as the tile tasks left them."""
    rng = np.random.default_rng(5)
    n, d = 700, 2
    Xc = rng.standard_normal((n, d))
    cat = rng.integers(0, 3, size=n).astype(float)
    X = np.vstack([np.column_stack([Xc, cat, np.zeros(n)]), np.column_stack([Xc, cat, np.ones(n)])])
    y = np.sin(X[:, 0]) + 0.3 * X[:, 3] + 0.1 * rng.standard_normal(2 * n)
    spec = golden_spec("composite_N140")  # the golden case's model (and its hyper-parameters) on a table ten times the size
    theta = GOLD["composite_N140/theta"]
    val_r, grad_r = O.nlml_and_grad(spec, theta, X, y)
    eng = make_engine(spec, theta, X, y)
    for scheme in (eng.GRAD_LAUNCH_TREE, eng.GRAD_FUSED):
        eng.set_grad_scheme(scheme)
        val, g = eng.evaluate(theta)
        assert abs(val - val_r) < 1e-10 * max(1.0, abs(val_r)) and rel(g, grad_r) < 1e-8
    eng.close()


@pytest.mark.parametrize("N", [300, 3000])
def test_fused_evaluation_equals_the_three_calls(gpu, N):
    """gmb_evaluate = gmb_set_theta + gmb_factorize + gmb_nlml with the gradient enqueued right behind the factorisation: the
    same bits, the factor stays usable for a prediction, and a covariance that is not positive definite is reported as by
    gmb_factorize (the gradient that ran on the failed factor is dropped)."""
    d = 3
    X, y, ls = O.synthetic_table(N, d, seed=17)
    spec = O.make_spec(d, range(d), kind="Matern32")
    theta = O.pack_theta(spec, ls, 1.2, 0.3)
    Xs = np.random.default_rng(1).standard_normal((64, d))
    eng = make_engine(spec, theta, X, y)
    eng.factorize()
    val_a, g_a = eng.nlml(grad=True)
    mu_a, var_a = eng.predict(Xs)
    val_b, g_b = eng.evaluate(theta * 1.0)
    assert eng.factor_is_current()
    mu_b, var_b = eng.predict(Xs)
    if N >= 768:  # both routes run the tile schedules: the same bits
        assert np.float64(val_a).tobytes() == np.float64(val_b).tobytes() and g_a.tobytes() == g_b.tobytes()
        assert mu_a.tobytes() == mu_b.tobytes() and var_a.tobytes() == var_b.tobytes()
    else:  # two to five block columns: gmb_evaluate is the fused tile launch, gmb_factorize the recursion -- same numbers, other grouping of the sums
        assert abs(val_a - val_b) < 1e-12 * abs(val_a) and rel(g_b, g_a) < 1e-10
        assert rel(mu_b, mu_a) < 1e-11 and np.max(np.abs(var_b - var_a)) < 1e-12
    assert np.float64(eng.evaluate(theta, grad=False)).tobytes() == np.float64(val_a).tobytes()
    val_r, grad_r = O.nlml_and_grad(spec, theta, X, y)
    assert abs(val_b - val_r) < 1e-10 * max(1.0, abs(val_r)) and rel(g_b, grad_r) < 1e-8
    Xbad = X.copy()
    Xbad[N // 2, 1] = np.nan
    eng.set_data(Xbad, y)
    with pytest.raises(np.linalg.LinAlgError):
        eng.evaluate(theta)
    assert eng.notpd_index() >= 0 and not eng.factor_is_current()
    eng.set_data(X, y)
    assert np.float64(eng.evaluate(theta)[0]).tobytes() == np.float64(val_b).tobytes()
    eng.close()

