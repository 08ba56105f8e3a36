"""C-ABI library: builds, loads and exports every symbol include/gumbi_hip.h declares.  No compute
call is made here (no GPU in the build container); the product path must refuse to run without
a device instead of falling back to the CPU."""
import re
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


@pytest.fixture(scope="module")
def lib():
    from gumbi_amd import build, engine

    build.build_library()
    return engine.load_library()


def test_header_symbols_are_all_exported(lib):
    from gumbi_amd import engine

    header = (ROOT / "include" / "gumbi_hip.h").read_text()
    declared = set(re.findall(r"\b(gmb_[a-z_0-9]+)\s*\(", header))
    assert declared, "no prototypes parsed from the header"
    assert declared == set(engine.exported_symbols()), declared ^ set(engine.exported_symbols())
    for name in declared:
        assert hasattr(lib, name), f"{name} missing from libgumbi_hip.so"
    header = (ROOT / "include" / "gumbi_hip.h").read_text()
    declared = int(re.search(r"#define\s+GMB_ABI_VERSION\s+(\d+)", header).group(1))
    assert lib.gmb_abi_version() == declared == engine.ABI_VERSION


def test_struct_layouts_match_header():
    from gumbi_amd import engine

    # gmb_kernel_spec: 3 + 16 + 1 + 8 + 1 + 4 + 4 + 4 int32 (= 41), 4 bytes of padding, then one double
    assert engine.C.sizeof(engine._Spec) == 41 * 4 + 4 + 8
    # gmb_timings: every member is 8 bytes (double / int64_t) -- count them in the header, and compare names
    header = (ROOT / "include" / "gumbi_hip.h").read_text()
    body = re.search(r"typedef struct gmb_timings \{(.*?)\} gmb_timings;", header, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = re.findall(r"\b(?:double|int64_t)\s+([a-z_0-9]+)\s*;", body)
    assert fields == [n for n, _ in engine.Timings._fields_]
    assert engine.C.sizeof(engine.Timings) == 8 * len(fields) == 56 * 8
    # gmb_comm {int32 rank, world; void* ctx; fn* all_gather} and gmb_dist_step {9 x int32, pad, int64}
    assert engine.C.sizeof(engine.GmbComm) == 24 and engine.C.sizeof(engine.DistStep) == 48
    spec = engine.KernelSpec(D=6, idx_cont=[0, 1, 2], idx_lin=[1], coreg=[(3, 4)], out_col=5, n_out=2)
    cs = spec.to_c()
    assert (cs.n_cont, cs.n_lin, cs.n_coreg, cs.out_col, cs.n_out, cs.hetero_noise) == (3, 1, 1, 5, 2, 1)
    assert cs.jitter == 1e-6 and list(cs.idx_cont)[:3] == [0, 1, 2]
    assert spec.theta_size() == 3 + 2 + 2 + 12 + 6 + 6
    lib = engine.load_library()
    assert lib.gmb_theta_size(engine.C.byref(cs)) == spec.theta_size()
    with pytest.raises(ValueError):
        engine.KernelSpec(D=2, idx_cont=[0], kind="RatQuad").to_c()
    with pytest.raises(ValueError):
        engine.KernelSpec(D=20, idx_cont=list(range(17))).to_c()


def test_no_cpu_fallback_without_device(lib):
    from gumbi_amd import engine

    if lib.gmb_device_count() > 0:
        pytest.skip("a GPU is visible here; the refusal path is exercised on CPU-only machines")
    assert lib.gmb_device_count() == engine.GMB_ENODEVICE
    with pytest.raises(engine.GumbiHipError, match="no CPU fallback"):
        engine.Engine()
    with pytest.raises(engine.GumbiHipError):
        engine.ls_limits(np.zeros((4, 2)), ard=False)


def test_product_package_never_imports_oracle():
    for path in (ROOT / "gumbi_amd").rglob("*.py"):
        text = path.read_text()
        assert "oracle" not in text.replace("no CPU fallback", ""), f"{path} mentions the oracle"
    for path in (ROOT / "gumbi_amd" / "csrc").glob("*"):
        if path.suffix in (".hip", ".hpp"):
            assert "oracle/" not in path.read_text().replace("see oracle/gp_oracle.py square_dist", "")


def _expected_tiles(mt, nt, bm, bn, tri, tri_off, stride):
    """Tiles the launch must compute: all of them, or -- tri -- those not strictly above the diagonal: tile
    (tm, tn) is skipped when its last row noff(tn) + bn - 1 + tri_off lies above its first column tm * bm."""
    out = set()
    for tm in range(mt):
        for tn in range(nt):
            e = tn * bn
            noff = (e >> 7) * stride * 128 + (e & 127)
            if not tri or noff + bn - 1 + tri_off >= tm * bm:
                out.add((tm, tn))
    return out


@pytest.mark.parametrize("strip", [0, 8, 4, 3])
def test_gemm_tile_lists_cover_every_tile_exactly_once(strip):
    """The kernels find their tile from the block index alone (XCD-balanced runs, triangular skipping, strided
    block rows, L2-aware strips): enumerate the launch on the host exactly as the kernel does and compare with
    the definition -- every computed tile once, no other tile, runs balanced."""
    from gumbi_amd import engine

    rng = np.random.default_rng(7)
    cases = [(1, 1, 128, 128, 0, 0, 1), (5, 7, 128, 128, 0, 0, 1), (16, 16, 128, 128, 1, 0, 1), (16, 40, 128, 128, 1, 8 * 128, 1),
             (24, 367, 128, 128, 1, 0, 1), (13, 29, 64, 64, 1, 0, 1), (8, 33, 128, 64, 1, 128, 1), (9, 50, 128, 32, 1, 0, 1),
             (40, 12, 128, 128, 1, 3 * 128, 8), (48, 13, 128, 128, 1, 5 * 128, 3), (31, 6, 64, 64, 1, 128, 2)]
    for _ in range(12):
        bm, bn = [(128, 128), (64, 64), (128, 64)][rng.integers(3)]
        cases.append((int(rng.integers(1, 60)), int(rng.integers(1, 90)), bm, bn, int(rng.integers(2)),
                      int(rng.integers(0, 6)) * 128, int(rng.integers(1, 5))))
    for mt, nt, bm, bn, tri, tri_off, stride in cases:
        grid, tiles = engine.gemm_tile_list(mt, nt, bm, bn, k=2048, tri=tri, tri_off=tri_off, nblk_stride=stride, strip=strip)
        got = [(tm, tn) for _, tm, tn in tiles]
        want = _expected_tiles(mt, nt, bm, bn, tri, tri_off, stride)
        assert len(got) == len(set(got)), (mt, nt, bm, bn, tri, tri_off, stride)
        assert set(got) == want, (mt, nt, bm, bn, tri, tri_off, stride, sorted(want - set(got))[:5], sorted(set(got) - want)[:5])
        assert grid % 8 == 0 and grid >= len(got)
        per_xcd = np.bincount([b % 8 for b, _, _ in tiles], minlength=8)
        if len(got) >= 64:  # equal work per XCD run (uniform k here): counts within a couple of tiles
            assert per_xcd.max() - per_xcd.min() <= max(3, strip + 1), per_xcd
        if strip and len(got) >= 8 * 64 and tri == 0:
            # locality: 64 consecutive tiles of one XCD's run touch few distinct A / B panels
            run0 = [(tm, tn) for b, tm, tn in tiles if b % 8 == 0][:64]
            assert len({tm for tm, _ in run0}) + len({tn for _, tn in run0}) <= strip + 64 // strip + 2 * 8


def test_gemm_longest_first_orders_cover_every_tile():
    from gumbi_amd import engine

    for order in (1, 2):
        for mt, nt, tri in [(12, 12, 1), (20, 9, 0), (7, 30, 1)]:
            grid, tiles = engine.gemm_tile_list(mt, nt, 128, 128, k=nt * 128, tri=tri, klo_n=int(order == 1),
                                                khi_n=int(order == 2), order=order)
            got = [(tm, tn) for _, tm, tn in tiles]
            # n-major lists: with tri the m-tiles of n-tile tn are [0, (tn*bn + bn - 1)/bm]
            want = {(tm, tn) for tn in range(nt) for tm in range(mt) if not tri or tm <= (tn * 128 + 127) // 128}
            assert len(got) == len(set(got)) and set(got) == want


@pytest.mark.parametrize("ti,tj", [(1, 1), (2, 1), (2, 2), (6, 5), (79, 79), (80, 79), (392, 391)])
@pytest.mark.parametrize("strip", [1, 2, 3, 4, 8])
def test_covariance_grid_covers_every_tile_exactly_once(ti, tj, strip):
    """cov_tile_kernel's enumerations (closed-form decode of the triangular grid, one rank's block rows of the
    multi-GPU build, the XCD-remapped rectangle of the cross-covariance), replayed on the host: every tile once,
    every strip inside one tile row, no workgroup beyond the grid."""
    from gumbi_amd.engine import cov_grid

    want_tri = {(i, j) for i in range(ti) for j in range(min(i + 1, tj))}
    grid, t = cov_grid(ti, tj, strip, tri_grid=1)
    tiles = list(map(tuple, t[:, 1:]))
    assert len(tiles) == len(set(tiles)) and set(tiles) == want_tri
    assert t[:, 0].max() == grid - 1 and len(np.unique(t[:, 0])) == grid      # no idle workgroup in the triangle
    for b in np.unique(t[:, 0])[:: max(1, grid // 50)]:
        rows = t[t[:, 0] == b]
        assert len(np.unique(rows[:, 1])) == 1 and len(rows) <= strip and np.all(np.diff(rows[:, 2]) == 1)
    for world in (1, 2, 3, 8):
        seen = []
        for rank in range(world):
            grid, t = cov_grid(ti, tj, strip, tri_grid=0, row_first=rank, row_stride=world)
            assert np.all(t[:, 1] % world == rank) and (len(t) == 0 or t[:, 0].max() < grid)
            seen += list(map(tuple, t[:, 1:]))
        assert len(seen) == len(set(seen)) and set(seen) == want_tri
    for keep in (0, 1):
        grid, t = cov_grid(ti, tj, strip, tri_grid=0, keep_order=keep)
        tiles = list(map(tuple, t[:, 1:]))
        assert len(tiles) == len(set(tiles)) == ti * tj and grid == ti * -(-tj // strip)


def test_tile_cholesky_tickets_enumerate_every_tile_once_in_a_topological_order():
    from gumbi_amd import engine

    """The ticket -> tile map the persistent tile Cholesky and the host share (gmb_debug_chol_task): every tile of the lower
    block triangle exactly once, and every tile after the tiles it waits for -- (I, k), (J, k) for k < J and (J, J) --
    which is what makes the launch deadlock-free whatever the residency of its workgroups."""
    for nct, nrt in [(1, 1), (1, 2), (2, 2), (3, 4), (16, 16), (40, 41), (79, 79), (80, 81), (160, 161)]:
        n = engine.chol_task_count(nct, nrt)
        assert n == nct * nrt - nct * (nct - 1) // 2
        tiles = [engine.chol_task(t, nct, nrt) for t in range(n)]
        pos = {tile: t for t, tile in enumerate(tiles)}
        assert len(pos) == n and all(j <= i < nrt and 0 <= j < nct for i, j in tiles)
        step = max(1, n // 400)
        for t in range(0, n, step):
            i, j = tiles[t]
            deps = [(j, j)] if i != j else []
            deps += [(i, k) for k in range(j)] + [(j, k) for k in range(j)]
            assert all(pos[dep] < t for dep in deps)
    assert engine.chol_task(10 ** 6, 3, 4) == (-1, -1)


def test_evaluation_task_list_is_a_topological_order_of_its_dependency_graph():
    """The host-built ticket list of the persistent evaluation launch (csrc/eval_tiles.hpp, gmb_debug_eval_tasks): every Cholesky
    tile, every tile of U = L^-T, every tile of Sigma^-1 and every block row's alpha exactly once, and every task AFTER the tasks it waits for -- a
    task then only ever waits for smaller tickets, which are finished or held by a running workgroup: the launch cannot
    deadlock whatever the residency of its workgroups.  With and without the factorisation's own tasks, several lags."""
    from gumbi_amd import engine

    CHOL, INV, ZZ, FIN = 0, 1, 2, 3
    for nct, nrt, lag in [(1, 1, 1), (1, 2, 0), (2, 2, 1), (3, 4, 5), (6, 6, 2), (20, 21, 5), (41, 41, 10), (79, 79, 19), (80, 81, 24)]:
        for with_chol in (True, False):
            tasks = engine.eval_task_list(nct, nrt, with_chol, lag)
            pos = {t: i for i, t in enumerate(tasks)}
            assert len(pos) == len(tasks)
            n_tri = nct * (nct + 1) // 2
            assert sum(1 for k, _, _ in tasks if k == INV) == n_tri and sum(1 for k, _, _ in tasks if k == ZZ) == n_tri
            assert sum(1 for k, _, _ in tasks if k == CHOL) == (nct * nrt - nct * (nct - 1) // 2 if with_chol else 0)
            assert sorted(i for k, i, _ in tasks if k == FIN) == list(range(nct))
            assert all((k == CHOL and j <= i < nrt and j < nct) or (k == INV and i <= j < nct) or (k == ZZ and j <= i < nct) or
                       (k == FIN and i < nct and j == 0) for k, i, j in tasks)
            yb = nrt - 1 if nrt > nct else nct - 1  # the block row that holds row N
            step = max(1, len(tasks) // 500)
            for t in range(0, len(tasks), step):
                k, i, j = tasks[t]
                if k == CHOL:
                    deps = ([(CHOL, j, j)] if i != j else []) + [(CHOL, i, q) for q in range(j)] + [(CHOL, j, q) for q in range(j)]
                elif k == INV:  # tile (r, c) = (i, j): its own row up to c, block row c of L, the tile of column c with row N
                    deps = [(INV, i, q) for q in range(i, j)]
                    if with_chol:
                        deps += [(CHOL, j, q) for q in range(j + 1)] + [(CHOL, yb, j)]
                elif k == ZZ:  # Sigma^-1 (I, J): rows I and J of U from column I on
                    deps = [(INV, i, q) for q in range(i, nct)] + [(INV, j, q) for q in range(i, nct)]
                else:  # alpha of block row i: the parts every tile of U's row i left (and, through them, row N of the factor)
                    deps = [(INV, i, q) for q in range(i, nct)]
                assert all(pos[dep] < t for dep in deps), (nct, nrt, lag, with_chol, tasks[t])



def test_paired_sigma_inverse_tasks_cover_every_tile_once_and_keep_the_order_topological():
    """The evaluation launch's list for large matrices (gmb_set_eval_pairs; csrc/eval_tiles.hpp: et_zz2_task): a Sigma^-1 task word
    with 0x4000 in its J field computes the tiles (I, J) and (I, J + 1) in one pass over block row I of U.  Every tile of the lower
    block triangle exactly once, pairs only below the diagonal's left neighbour (J + 1 <= I), everything else as in the plain
    list and in the same order, and every Sigma^-1 task behind all the tiles of U it reads."""
    from gumbi_amd import engine

    CHOL, INV, ZZ, FIN = 0, 1, 2, 3
    PAIR = 0x4000
    for nct, nrt, lag in [(1, 1, 0), (2, 3, 1), (7, 7, 2), (72, 73, 18), (157, 157, 24)]:
        for with_chol in (True, False):
            plain = engine.eval_task_list(nct, nrt, with_chol, lag)
            paired = engine.eval_task_list(nct, nrt, with_chol, lag, pairs=True)
            assert [t for t in paired if t[0] != ZZ] == [t for t in plain if t[0] != ZZ]
            tiles = []
            for k, i, j in paired:
                if k != ZZ:
                    continue
                if j & PAIR:
                    j &= ~PAIR
                    assert j + 1 <= i < nct
                    tiles += [(i, j), (i, j + 1)]
                else:
                    assert j <= i < nct
                    tiles.append((i, j))
            assert tiles == [(i, j) for k, i, j in plain if k == ZZ]  # the same tiles in the same order, two at a time
            n_pairs = sum(1 for k, _, j in paired if k == ZZ and j & PAIR)
            assert n_pairs == sum((i + 1) // 2 for i in range(nct))
            first_zz = min(t for t, (k, _, _) in enumerate(paired) if k == ZZ)
            assert all(k != INV for k, _, _ in paired[first_zz:])  # Sigma^-1 only reads U: every INV task has a smaller ticket
