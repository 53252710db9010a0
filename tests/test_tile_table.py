"""The measured tile table compiled into gemm.hip (csrc/tile_table.inc) and the tool that writes it (CPU checks: the table is
generated on a GPU box by tools/fwd_tune.py -> tools/gen_tile_table.py, so a malformed row would only show up as a wrong launch)."""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INC = os.path.join(ROOT, "pnpinversion_amd", "csrc", "tile_table.inc")
TILE_IDS = {0, 1, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16,  17, 18}       # launch_igemm's configuration ids (2 is force-only; 16 / 17: the 8-wave ping-pong kernel)
SPLITS = {1, 2, 3, 4, 6, 8, 12, 16}                                 # the split-K factors launch_igemm's cost model also walks
ROW = re.compile(r"^\{(\d+), (\d+), (\d+), (\d+), (\d+), (\d+)\},")


def _rows(path):
    out = []
    for line in open(path):
        m = ROW.match(line)
        if m:
            out.append(tuple(int(g) for g in m.groups()))
        else:
            assert line.startswith("//") or not line.strip(), line
    return out


def test_committed_table_is_well_formed():
    rows = _rows(INC)
    assert len(rows) >= 100
    keys = [r[:4] for r in rows]
    assert len(set(keys)) == len(keys), "duplicate {M, N, K, ksize}: the first match would shadow the second"
    for M, N, K, ks, cfg, split in rows:
        assert ks in (1, 3) and K % (ks * ks * 8) == 0 and M > 0 and N > 0
        assert cfg in TILE_IDS and split in SPLITS
        # a split must leave each slice at least one 64-wide k-chunk
        assert split == 1 or K // 64 >= split
    # the row counts of the benchmarked schedules are covered: 64 x 64 level of the 1-, 3-, 12- and 96-row launches
    for rows_per_launch in (1, 3, 12, 96):
        assert (rows_per_launch * 4096, 320, 2880, 3) in set(keys)


def test_gen_tile_table_picks_the_fastest_and_prefers_simple_within_two_percent(tmp_path):
    src = {"rows": 12, "shapes": [
        {"M": 49152, "N": 320, "K": 2880, "ks": 3, "us": {"auto": 1.0, "128": 100.0, "320": 90.0, "64/s2": 95.0}},     # wide tile wins
        {"M": 3072, "N": 1280, "K": 11520, "ks": 3, "us": {"128": 101.0, "256n/s4": 100.0, "unknown_cfg": 1.0}},       # 1 %: keep the plain 128
        {"M": 768, "N": 1280, "K": 1280, "ks": 1, "us": {"64": 30.0, "64k4": 20.0}}]}
    j = tmp_path / "t.json"
    j.write_text(json.dumps(src))
    out = tmp_path / "t.inc"
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_tile_table.py"), str(out), str(j)], check=True, capture_output=True)
    assert _rows(out) == [(768, 1280, 1280, 1, 8, 1), (3072, 1280, 11520, 3, 0, 1), (49152, 320, 2880, 3, 4, 1)]


def test_runtime_lookup_exact_and_nearest_row_count():
    """pnpi_tile_table_lookup (host-only; the function launch_igemm itself calls): every committed row is found exactly, the same layer at an
    untabled row count (--batch_size 2 ... 7: 24 ... 84 rows) resolves to the nearest tabled M within 4x, anything else to the cost model."""
    import ctypes as C
    from pnpinversion_amd import _capi
    lib = _capi.load_library()
    cfg, split, em = C.c_int(-1), C.c_int(-1), C.c_int(-1)
    rows = _rows(INC)
    for (M, N, K, ks, c, s) in rows:
        assert lib.pnpi_tile_table_lookup(M, N, K, ks, C.byref(cfg), C.byref(split), C.byref(em)) == 1
        assert (cfg.value, split.value, em.value) == (c, s, M)
    by_layer = {}
    for r in rows:
        by_layer.setdefault(r[1:4], []).append(r)
    # the 320 -> 320 3x3 convs (64 x 64 level: M = rows * 4096; its stride-2 downsampler: M = rows * 1024) at 20 and 60 rows: the entry
    # nearest in ratio is the 96-row downsampler (98304 = 1.2x) resp. the 96-row level-0 conv (393216 = 1.6x)
    tabled = sorted(r[0] for r in by_layer[(320, 2880, 3)])
    for M in (20 * 4096, 60 * 4096):
        assert M not in tabled
        assert lib.pnpi_tile_table_lookup(M, 320, 2880, 3, C.byref(cfg), C.byref(split), C.byref(em)) == 2
        nearest = min(tabled, key=lambda m: max(m / M, M / m))
        assert em.value == nearest and max(nearest / M, M / nearest) <= 4
        want = next(r for r in by_layer[(320, 2880, 3)] if r[0] == nearest)
        assert (cfg.value, split.value) == want[4:]
    assert lib.pnpi_tile_table_lookup(20 * 4096, 320, 2880, 3, None, None, C.byref(em)) == 2 and em.value == 96 * 1024
    assert lib.pnpi_tile_table_lookup(60 * 4096, 320, 2880, 3, None, None, C.byref(em)) == 2 and em.value == 96 * 4096
    # beyond 4x of every entry, another layer, another kernel size, nonsense: no entry
    big = max(r[0] for r in by_layer[(320, 2880, 3)])
    assert lib.pnpi_tile_table_lookup(5 * big, 320, 2880, 3, None, None, None) == 0
    assert lib.pnpi_tile_table_lookup(4096, 328, 2880, 3, None, None, None) == 0
    assert lib.pnpi_tile_table_lookup(4096, 320, 2880, 1, None, None, None) == 0
    assert lib.pnpi_tile_table_lookup(0, 320, 2880, 3, None, None, None) == 0
    # the nearest-entry rule can be switched off (A/B knob)
    assert lib.pnpi_set_tuning(b"igemm_table_near", 0) == 0
    try:
        assert lib.pnpi_tile_table_lookup(20 * 4096, 320, 2880, 3, None, None, None) == 0
        assert lib.pnpi_tile_table_lookup(12 * 4096, 320, 2880, 3, None, None, None) == 1
    finally:
        assert lib.pnpi_set_tuning(b"igemm_table_near", 1) == 0
