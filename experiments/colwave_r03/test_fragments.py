"""experiments/colwave_r03/test_fragments.py -- the CPU tests of the bracket path and of the EDT_CONTIG block form as they stood in
tests/test_lane_logic.py at the end of round 3 (they need the emulator fragments next to this file).  NOT collected."""

@pytest.fixture(scope="module")
def emul_contig():
    """the same emulation built with -DEDT_CONTIG: the windowed path as one call per block with the block's position an
    argument (brute_block; the experiment of DESIGN.md 7.1, not the shipped form)"""
    os.makedirs(BUILD, exist_ok=True)
    so = os.path.join(BUILD, "liblane_emul_contig.so")
    src = os.path.join(ROOT, "tests", "lane_emul.cpp")
    hdr = os.path.join(CSRC, "edt_colwave_lane.h")
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        tmp = f"{so}.{os.getpid()}.tmp"
        subprocess.run(["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-fno-fast-math", "-shared", "-DEDT_CONTIG",
                        "-fPIC", f"-I{CSRC}", src, "-o", tmp], check=True)
        os.replace(tmp, so)
    lib = ctypes.CDLL(so)
    lib.lane_emul_column_pass_mode.restype = ctypes.c_int
    return lib


@pytest.mark.parametrize("mode", ["window", "window64", "window_even", "auto"])
@pytest.mark.parametrize("n,sx,kind", [c for c in CASES if c[0] in (1025, 513, 300, 257, 130, 64, 33, 17) and (c[2] != "ones" or c[0] <= 300)])
def test_block_form_of_the_windowed_path(emul_contig, oracle_port, n, sx, kind, mode):
    """brute_block (EDT_CONTIG): every block of every band through the per-block entry, against the oracle"""
    rng = np.random.default_rng(n * 31 + sx)
    lab = make_labels(n, sx, kind, rng)
    for (wx, wy) in ((1.0, 1.0), (6.0, 30.0), (0.7, 1.3)):
        for bb in (True, False):
            f1 = x_pass(oracle_port, lab, wx, bb)
            want = oracle_port.raw2d(lab, 2, sx, n, (wx, wy), bb).reshape(n, sx)
            got = column_pass(emul_contig, lab, f1, wy, bb, 0 if bb else 1, MODES[mode])
            ev = slice(None, None, 2) if mode.endswith("_even") else slice(None)
            assert np.array_equal(got[ev], want[ev]), (n, sx, kind, wx, wy, bb)
            if mode.endswith("_even"):
                assert np.array_equal(got[1::2], f1[1::2])
    # fp32 fma candidates (mode 7) through the same entry
    emul_contig.lane_emul_set_fmin(ctypes.c_float(float(np.float32(3.58) * np.float32(3.58))), ctypes.c_int(1024))
    f1 = x_pass(oracle_port, lab, 3.58, True)
    got = column_pass(emul_contig, lab, f1, 40.0, True, 0, 7)
    assert np.array_equal(got, oracle_port.raw2d(lab, 2, sx, n, (3.58, 40.0), True).reshape(n, sx))






@pytest.mark.parametrize("seed", range(12))
def test_bracket_path_on_arbitrary_fields(emul, seed):
    """The bracket path (mono_band) against the hull path (pinned to the oracle above) on fields that are NOT the
    output of a pass over the same labels: random integers, smooth bowls with spikes, plateaus with ties, fields with
    huge dynamic range -- whatever makes brackets wide, anchors' windows short (borders close by) and argmins tie."""
    rng = np.random.default_rng(9000 + seed)
    for t in range(14):
        n = int(rng.choice([33, 64, 100, 257, 300, 512, 700, 1024]))
        sx = int(rng.choice([3, 16, 32, 37]))
        kind = t % 7
        yy = np.arange(n, dtype=np.float64)[:, None]
        if kind == 0:
            f = rng.integers(0, 5000, size=(n, sx)).astype(np.float32)
        elif kind == 1:   # bowls: far owners, long windows
            c = rng.uniform(0, n, size=(1, sx))
            f = np.floor((yy - c) ** 2 * rng.uniform(0.2, 3.0)) + rng.integers(0, 3, size=(n, sx))
        elif kind == 2:   # plateaus: ties everywhere
            f = np.repeat(rng.integers(0, 40, size=(-(-n // 16), sx)), 16, axis=0)[:n].astype(np.float64) ** 2
        elif kind == 3:   # a few deep wells in a high field
            f = np.full((n, sx), 250000.0)
            f[rng.random((n, sx)) < 0.02] = 0.0
            f += rng.integers(0, 2, size=(n, sx))
        elif kind == 4:   # ramps (the x-distance field of slanted cell walls)
            f = (np.abs(yy * rng.uniform(-1.5, 1.5, size=(1, sx)) + rng.uniform(-200, 200, size=(1, sx))).astype(np.int64) % 180).astype(np.float64) ** 2
        elif kind == 5:   # huge dynamic range, still exact
            f = (rng.integers(0, 2000, size=(n, sx)) ** 2).astype(np.float64)
        else:
            f = rng.integers(0, 30, size=(n, sx)).astype(np.float64)
        f = np.asarray(f, dtype=np.float32)
        lab = make_labels(n, sx, ["blocky", "noise", "membrane", "ones"][int(rng.integers(0, 4))], rng)
        if t % 3 == 0:
            lab[:] = 1   # one run per column: everything hinges on the brackets
        f[lab == 0] = 0.0
        for w in (1.0, 2.0, 30.0, 0.5):
            for bb in (True, False):
                want = column_pass(emul, lab, f, w, bb, 0, MODES["hull"])
                got = column_pass(emul, lab, f, w, bb, 0, MODES["mono"])
                assert np.array_equal(got, want), (seed, t, n, sx, kind, w, bb, np.argwhere(got != want)[:4])
    emul.lane_emul_mono_tiles.restype = ctypes.c_long
    assert emul.lane_emul_mono_tiles() > 100   # most of these tiles really took the bracket path
