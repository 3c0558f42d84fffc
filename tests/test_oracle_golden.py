"""Pins the CPU oracle (oracle/psac_ref.cpp) against the reference's own
known-answer vectors and the checksums captured from the reference
(tests/golden/reference_kat.json; provenance inside)."""
import json
import os

import numpy as np
import pytest

import inputs
import oracle_lib as O

HERE = os.path.dirname(os.path.abspath(__file__))
KAT = json.load(open(os.path.join(HERE, "golden", "reference_kat.json")))


def make_input(row):
    g = row["gen"]
    if g == "literal":
        return O.as_text(row["arg"])
    if g == "rand_dna":
        return O.rand_dna(row["n"], row["seed"])
    if g == "cyclic":
        return inputs.cyclic(row["n"], row["arg"])
    if g == "tandem_rand_dna":
        return inputs.tandem(row["n"], row["period"], O.rand_dna(row["period"], row["seed"]))
    if g == "splitmix_mod127p1":
        return inputs.bytes_mod127p1(row["n"], row["seed"])
    if g == "splitmix_dna":
        return inputs.dna(row["n"], row["seed"])
    if g == "splitmix_ascii128":
        return inputs.ascii128(row["n"], row["seed"])
    raise KeyError(g)


def test_mississippi_exact():
    m = KAT["mississippi"]
    for bits in (32, 64):
        for fast in (True, False):
            r = O.construct(m["text"], bits=bits, fast=fast)
            assert r["SA"].tolist() == m["SA"]
            assert r["ISA"].tolist() == m["ISA"]
            assert r["LCP"].tolist() == m["LCP"]


def test_int_alphabet_mississippi_order():
    # test/test_psac.cpp:277-304 uses an int alphabet; the byte engine sees the
    # same order after densifying the symbols.
    m = KAT["int_alphabet_mississippi"]
    sym = sorted(set(m["text"]))
    dense = bytes(1 + sym.index(c) for c in m["text"])
    assert O.construct(dense, bits=32)["SA"].tolist() == m["SA"]


def test_bitops_kats():
    b = KAT["bitops"]
    for x, bits, exp in b["trailing_zeros"]:
        assert O.trailing_zeros(int(x, 16), bits) == exp
    for x, bits, exp in b["leading_zeros"]:
        assert O.leading_zeros(int(x, 16), bits) == exp
    for x, y, bits, k, l, exp in b["lcp_bitwise"]:
        assert O.lcp_bitwise(int(x, 16), int(y, 16), bits, k, l) == exp
    for x, exp in b["floorlog2"]:
        assert O.floorlog2(int(x, 16)) == exp
    for x, exp in b["ceillog2"]:
        assert O.ceillog2(int(x, 16)) == exp


def test_bitops_random_against_naive():
    # test/test_bitops.cpp:36-48, 62-74: 100000 random words vs bit loops (10000 here)
    rng = np.random.default_rng(5)
    for x in rng.integers(1, 2**63, size=10000, dtype=np.uint64).tolist():
        assert O.leading_zeros(x, 64) == 64 - x.bit_length()
        assert O.trailing_zeros(x, 64) == (x & -x).bit_length() - 1


@pytest.mark.parametrize("row", KAT["checksums"], ids=[r["name"] for r in KAT["checksums"]])
def test_reference_checksums(row):
    text = make_input(row)
    assert text.size == row["n"]
    bits = 64 if row["name"] in ("rand_dna_66763_23", "ascii128_1M_42", "bytes127_1M_42") else 32
    r = O.construct(text, bits=bits)
    assert "%016x" % O.fnv(r["SA"]) == row["sa"]
    assert "%016x" % O.fnv(r["ISA"]) == row["isa"]
    assert "%016x" % O.fnv(r["LCP"]) == row["lcp"]
    assert int(r["LCP"].max()) == row["max_lcp"]
    assert int(r["LCP"].astype(np.uint64).sum()) == row["sum_lcp"]
    # independent checks: order property + Kasai (check_suffix_array.hpp:56-88, lcp.hpp:46-77)
    assert O.check_sa(text, r["SA"], r["ISA"]) == 0
    assert np.array_equal(O.kasai(text, r["SA"], r["ISA"]), r["LCP"])


def test_tandem_round_trace():
    t = KAT["tandem_trace"]
    row = [r for r in KAT["checksums"] if r["name"] == "tandem_1M_1024_3"][0]
    text = make_input(row)
    r = O.construct(text, bits=32)
    assert r["k"] == t["k"]
    doubling = [x for x in r["trace"] if x[3] == 0]
    assert len(doubling) == t["rounds"]
    n = t["n"]
    for i, (h, ub, ue, _) in enumerate(doubling):
        assert h == t["k"] << i
        if i < t["rounds"] - 1:
            assert ub == 1024 and ue == n - 2 * h + 1
        else:
            assert ub == 0 and ue == 0


@pytest.mark.parametrize("bits,fast,k", [(32, True, 0), (32, True, 3), (32, False, 2), (64, True, 0), (64, True, 3)])
def test_variants_agree_with_naive(bits, fast, k):
    # test/test_psac.cpp:131-176 (RandAll) and :250-274 (Lcp1): default k, k=3 (forces
    # bucket chasing), fast=false with k=2.
    text = O.rand_dna(20011, 7)
    r = O.construct(text, bits=bits, fast=fast, k=k)
    assert np.array_equal(r["SA"], O.naive_sa(text, bits))
    assert O.check_sa(text, r["SA"], r["ISA"]) == 0
    assert np.array_equal(O.kasai(text, r["SA"], r["ISA"]), r["LCP"])


def test_repeats_dictionary_words():
    # test/test_psac.cpp:178-224 (RepeatsAll): concatenated dictionary words
    words = ["helloworld", "blahlablah", "ellow", "worldblah", "rld", "hello"]
    rng = np.random.default_rng(3)
    s = "".join(words[i] for i in rng.integers(0, len(words), 3000))
    for k in (0, 3):
        r = O.construct(s, bits=64, k=k)
        assert np.array_equal(r["SA"], O.naive_sa(s, 64))
        assert np.array_equal(O.kasai(s, r["SA"], r["ISA"]), r["LCP"])


def test_small_strings():
    # test/test_psac.cpp:226-248 (n = 9) plus the degenerate sizes
    for n in (2, 3, 4, 5, 9, 17):
        text = O.rand_dna(n, 13)
        r = O.construct(text, bits=32)
        assert np.array_equal(r["SA"], O.naive_sa(text, 32)), n
        assert np.array_equal(O.kasai(text, r["SA"], r["ISA"]), r["LCP"]), n
    r = O.construct(b"A", bits=32)
    assert r["SA"].tolist() == [0] and r["ISA"].tolist() == [0] and r["LCP"].tolist() == [0]
    for s in (b"AAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAA", b"ABABABABABABABABABABABABABA", b"\x00\x00\x01\x00\x00"):
        r = O.construct(s, bits=32)
        assert np.array_equal(r["SA"], O.naive_sa(s, 32))
        assert np.array_equal(O.kasai(s, r["SA"], r["ISA"]), r["LCP"])


def test_ansv_definitions():
    # test/test_ansv.cpp:255-268 sizes; brute-force definitions
    rng = np.random.default_rng(11)
    for n in (8, 137, 1000):
        v = rng.integers(0, 20, n).astype(np.uint32)
        NO = 2**64 - 1
        for left in (True, False):
            sm = O.ansv(v, left, 0, NO); eq = O.ansv(v, left, 1, NO); fe = O.ansv(v, left, 2, NO)
            assert np.array_equal(sm, O.ansv_seq(v, left, NO))
            for i in range(n):
                rng_idx = range(i - 1, -1, -1) if left else range(i + 1, n)
                e_sm = e_eq = NO
                for j in rng_idx:
                    if e_eq == NO and v[j] <= v[i]:
                        e_eq = j
                    if v[j] < v[i]:
                        e_sm = j
                        break
                assert sm[i] == e_sm and eq[i] == e_eq
                e_fe = e_eq
                if e_eq != NO:
                    step = -1 if left else 1
                    j = e_eq + step
                    while 0 <= j < n and v[j] >= v[e_eq]:
                        if v[j] == v[e_eq]:
                            e_fe = j
                        j += step
                assert fe[i] == e_fe


def test_suffix_tree_mississippi_table():
    # test/test_suffixtree.cpp:68-83
    m = KAT["mississippi"]
    r = O.construct(m["text"], bits=64)
    nodes = O.suffix_tree(m["text"], r["SA"], r["LCP"])
    assert nodes.reshape(-1).tolist() == m["suffix_tree_nodes"]


def test_left_branching_chars():
    # suffix_array<char, T, true, true>: Lc carried through the range minima (suffix_array.hpp:1365-1383,
    # par_rmq.hpp:334-481) equals its definition Lc[i] = S[SA[i-1] + LCP[i]] (desa.hpp:262-264)
    m = KAT["mississippi"]
    r = O.construct_lc(m["text"], bits=64)
    assert r["SA"].tolist() == m["SA"] and r["LCP"].tolist() == m["LCP"]
    # by hand from SA = 10 7 4 1 0 9 8 6 3 5 2, LCP = 0 1 1 4 0 0 1 0 2 1 3: S[SA[i-1] + LCP[i]]
    assert bytes(r["Lc"]) == b"\x00\x00ppimippip"
    cases = [(O.rand_dna(5000, 3), 32, True, 0), (O.rand_dna(5000, 3), 64, False, 2), (inputs.cyclic(4000, "abc"), 32, True, 0),
             (inputs.tandem(6000, 64, inputs.dna(64, 5)), 32, True, 0), (inputs.ascii128(7000, 9), 64, True, 0),
             (inputs.bytes_mod127p1(3000, 4), 32, True, 3), (O.as_text("aaaaaaaaaaaaaaaa"), 32, True, 0),
             (O.as_text("ab"), 32, True, 0)]
    for text, bits, fast, k in cases:
        r = O.construct_lc(text, bits=bits, fast=fast, k=k)
        base = O.construct(text, bits=bits, fast=fast, k=k)
        assert np.array_equal(r["SA"], base["SA"]) and np.array_equal(r["LCP"], base["LCP"])
        assert np.array_equal(r["Lc"], O.left_chars_by_definition(text, r["SA"], r["LCP"]))


# ---- generalized suffix array (test/test_gsa.cpp)
def repeat_inc_seq(seq, reps):                       # test/test_gsa.cpp:27-33
    return [seq * (i + 1) for i in range(reps)]


def repeat_inc_gsa(slen, reps):                      # test/test_gsa.cpp:35-52
    m = reps * (reps + 1) // 2
    gsa = [0] * (slen * m)
    for i in range(slen):
        o = i * m
        for j in range(reps):
            gsa[o] = i + slen * (j * (j + 1)) // 2
            o += 1
            for k in range(j + 2, reps + 1):
                gsa[o] = gsa[o - 1] + k * slen
                o += 1
    return gsa


def repeat_inc_glcp(slen, reps):                     # test/test_gsa.cpp:54-71
    m = reps * (reps + 1) // 2
    lcp = [0] * (slen * m)
    for i in range(slen):
        o = i * m
        lcp[o] = 0
        o += 1
        for j in range(1, reps):
            for _ in range(reps + 1 - j):
                lcp[o] = j * slen - i
                o += 1
    return lcp


GSA_REPEATS = [("ab", 3), ("abc", 3), ("a", 20), ("abc", 10), ("abcdef", 50)]   # test/test_gsa.cpp:156-179


def test_gsa_simple_tiny():
    # test/test_gsa.cpp:73-105
    r = O.construct_ss(["abab", "baba"], bits=64)
    assert r["SA"].tolist() == [7, 2, 5, 0, 3, 6, 1, 4]
    assert r["LCP"].tolist() == [0, 1, 2, 3, 0, 1, 2, 3]


@pytest.mark.parametrize("seq,reps", GSA_REPEATS)
def test_gsa_inc_repeats(seq, reps):
    # test/test_gsa.cpp:107-179: strings seq, seq^2, ..., seq^reps
    for bits in (64, 32):
        r = O.construct_ss(repeat_inc_seq(seq, reps), bits=bits)
        assert r["SA"].tolist() == repeat_inc_gsa(len(seq), reps)
        assert r["LCP"].tolist() == repeat_inc_glcp(len(seq), reps)
        assert np.array_equal(r["ISA"][r["SA"].astype(np.int64)], np.arange(r["SA"].size, dtype=r["ISA"].dtype))


def test_gsa_random_sets_against_definition():
    rng = np.random.RandomState(5)
    for trial in range(12):
        m = int(rng.randint(1, 40))
        sigma = int(rng.choice([1, 2, 4, 20]))
        strings = [bytes(rng.randint(97, 97 + sigma, size=int(rng.randint(1, 60))).astype(np.uint8)) for _ in range(m)]
        for bits, k in ((32, 0), (64, 0), (64, 2)):
            r = O.construct_ss(strings, bits=bits, k=k)
            SA, LCP = O.gsa_by_definition(strings)
            assert np.array_equal(r["SA"].astype(np.uint64), SA), (trial, bits, k)
            assert np.array_equal(r["LCP"].astype(np.uint64), LCP), (trial, bits, k)


def test_all_cores_build_matches_scalar_build():
    # oracle/libpsac_oracle_mt.so (OpenMP loops + parallel-mode sort; bench.py's CPU baseline) against the scalar build
    for text in (O.rand_dna(200000, 4), inputs.tandem(50000, 100, O.rand_dna(100, 2)), inputs.ascii128(100000, 7)):
        SA, LCP = O.construct_all_cores(text, bits=32)
        ref = O.construct(text, bits=32)
        assert np.array_equal(SA, ref["SA"]) and np.array_equal(LCP, ref["LCP"])


# ---------------------------------------------------------------------------------------------------------------
# The oracle against the reference's own checker run here: libdivsufsort compiled from the sources under
# /root/reference/ext/libdivsufsort (oracle/Makefile -> oracle/_ref/), SA == dss::construct and LCP == Kasai on the
# shapes test/test_psac.cpp uses (:77-98, :50-74, :131-274), for both index widths and k / fast_resolval variants.
# ---------------------------------------------------------------------------------------------------------------
needs_dss = pytest.mark.skipif(not O.have_divsufsort(), reason="oracle/_ref/libdivsufsort*.so not built")


def _dss_cases():
    import inputs
    words = ["helloworld", "blahlablah", "ellow", "worldblah", "rld", "hello"]
    rng = np.random.default_rng(3)
    rep = "".join(words[i] for i in rng.integers(0, len(words), 15000)).encode()
    return [("mississippi", b"mississippi"), ("rand_dna_130370_7", O.rand_dna(130370, 7)),
            ("rand_dna_66763_23", O.rand_dna(66763, 23)), ("repeats", rep), ("abc13333", b"abc" * 13333),
            ("tandem_64", inputs.tandem(50000, 64, O.rand_dna(64, 1))), ("ascii128", inputs.ascii128(100000, 42)),
            ("all_bytes", bytes(range(256)) * 40), ("single_symbol", b"A" * 3000), ("n9", O.rand_dna(9, 13))]


@needs_dss
@pytest.mark.parametrize("name,text", _dss_cases(), ids=[c[0] for c in _dss_cases()])
def test_oracle_equals_divsufsort_and_kasai(name, text):
    for bits in (32, 64):
        SA, ISA, LCP = O.divsufsort_sa_lcp(text, bits)
        assert O.sufcheck(text, SA) == 0
        for fast, k in ((True, 0), (True, 3), (False, 2)):
            if len(text) < 30 and k:
                continue
            ref = O.construct(text, bits=bits, fast=fast, k=k)
            assert np.array_equal(ref["SA"], SA), (name, bits, fast, k)
            assert np.array_equal(ref["ISA"], ISA)
            assert np.array_equal(ref["LCP"], LCP)


@needs_dss
def test_divsufsort_reproduces_the_reference_known_answers():
    # the library build itself, against vectors the reference holds: mississippi (test/test_psac.cpp:105)
    m = KAT["mississippi"]
    assert O.divsufsort(m["text"], 32).tolist() == m["SA"]
    assert O.divsufsort(m["text"], 64).tolist() == m["SA"]
    # and against the survey-captured divsufsort + Kasai checksums
    for row in KAT["checksums"]:
        if row["n"] > (1 << 20):
            continue
        text = make_input(row)
        SA, ISA, LCP = O.divsufsort_sa_lcp(text, 64)
        assert "%016x" % O.fnv(SA) == row["sa"], row["name"]
        assert "%016x" % O.fnv(LCP) == row["lcp"], row["name"]


def build_dss_tool(name, out_dir, with_engine=False):
    """g++ of tests/cpp/<name>.cpp against oracle/_ref (libdivsufsort) and, for psac-vs-dss, libpsacx.so."""
    import subprocess
    root = os.path.dirname(HERE)
    ref = os.path.join(root, "oracle", "_ref")
    exe = os.path.join(str(out_dir), name)
    cmd = ["g++", "-std=c++11", "-O2", "-Wall", "-I" + os.path.join(ref, "include"), "-o", exe,
           os.path.join(HERE, "cpp", name + ".cpp"), os.path.join(ref, "libdivsufsort.so"), os.path.join(ref, "libdivsufsort64.so"),
           "-Wl,-rpath," + ref]
    if with_engine:
        lib = os.path.join(root, "psac_amd", "lib")
        cmd += ["-L" + lib, "-lpsacx", "-Wl,-rpath," + lib, "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.check_call(cmd)
    return exe


@needs_dss
def test_dss_command_line(tmp_path):
    # src/dss.cpp:41-84: one "<ms> ms" line per iteration; -f and -r exclude each other
    import subprocess
    exe = build_dss_tool("dss", tmp_path)
    r = subprocess.run([exe, "-r", "200000", "-s", "3", "-i", "2"], capture_output=True, text=True)
    assert r.returncode == 0 and len([l for l in r.stderr.splitlines() if l.endswith(" ms")]) == 2, r.stderr
    f = tmp_path / "m.txt"
    f.write_bytes(b"mississippi")
    assert subprocess.run([exe, "-f", str(f)], capture_output=True).returncode == 0
    assert subprocess.run([exe], capture_output=True).returncode != 0
    assert subprocess.run([exe, "-f", str(f), "-r", "5"], capture_output=True).returncode != 0
