"""Synthetic text generators."""
import numpy as np

_M = (1 << 64) - 1


def splitmix64_stream(n, seed):
    """n outputs of splitmix64 with state starting at `seed` (vectorised)."""
    with np.errstate(over="ignore"):
        idx = np.arange(1, n + 1, dtype=np.uint64)
        x = np.uint64(seed & _M) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = x
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def dna(n, seed):
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    return lut[(splitmix64_stream(n, seed) & np.uint64(3)).astype(np.int64)]


def ascii128(n, seed):
    return (splitmix64_stream(n, seed) & np.uint64(127)).astype(np.uint8)


def bytes_mod127p1(n, seed):
    return (1 + splitmix64_stream(n, seed) % np.uint64(127)).astype(np.uint8)


def tandem(n, period, unit):
    unit = np.asarray(unit, dtype=np.uint8)
    assert unit.size == period
    reps = (n + period - 1) // period
    return np.tile(unit, reps)[:n].copy()


def cyclic(n, word):
    w = np.frombuffer(word.encode(), dtype=np.uint8)
    return np.tile(w, (n + w.size - 1) // w.size)[:n].copy()


def mutated(n, period, seed):
    """TANDEM(n, period, seed) with one position in 200 given a character of its own (psacx_synth_text_dev kind 3):
    position g is mutated when the g-th output of the stream seeded seed ^ 0xA5A5A5A5A5A5A5A5 is a multiple of 200."""
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    t = tandem(n, period, dna(period, seed))
    m = splitmix64_stream(n, seed ^ 0xA5A5A5A5A5A5A5A5)
    hit = (m % np.uint64(200)) == 0
    t[hit] = lut[((m[hit] >> np.uint64(8)) & np.uint64(3)).astype(np.int64)]
    return t
