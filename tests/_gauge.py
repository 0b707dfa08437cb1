"""Helpers of the gauge-map tests (tests/test_oracle_kat.py, tests/test_gpu_couplings.py): a random gauge field g on the lattice,
the Mattis couplings J_ij = g_i g_j it induces as the reference's per-site nibbles <up, down, left, right> (optimized/main.cu:588-612:
a set bit flips that neighbour before the energy sum), and g itself packed like a colour's spins.  Colour-site k of row r sits in
lattice column 2k + (r & 1) for black, 2k + 1 - (r & 1) for white (:928-929)."""
import numpy as np


def _cols(X, Y, color):
    r = np.arange(Y)[:, None]
    k = np.arange(X // 2)[None, :]
    return 2 * k + ((r & 1) if color == 0 else 1 - (r & 1))


def _pack(nib, X, Y):
    nib = nib.astype(np.uint64).reshape(Y, X // 32, 16)
    sh = (np.arange(16, dtype=np.uint64) * np.uint64(4))[None, None, :]
    return (nib << sh).sum(axis=2, dtype=np.uint64)


def gauge_field(X, Y, seed):
    return np.random.default_rng(seed).integers(0, 2, size=(Y, X)).astype(np.uint8)  # 1 = this site's spin is flipped


def gauge_words(g, color):
    """g at the sites of `color`, packed like that colour's spin array: XOR it onto the spins."""
    Y, X = g.shape
    c = _cols(X, Y, color)
    return _pack(g[np.arange(Y)[:, None], c], X, Y)


def mattis_nibbles(g, color):
    """Bond nibbles of the sites of `color` for J_ij = g_i g_j (antiferromagnetic where the gauge differs), periodic lattice."""
    Y, X = g.shape
    c = _cols(X, Y, color)
    r = np.broadcast_to(np.arange(Y)[:, None], c.shape)
    me = g[r, c]
    nib = ((me ^ g[(r - 1) % Y, c]) << 3) | ((me ^ g[(r + 1) % Y, c]) << 2) | ((me ^ g[r, (c - 1) % X]) << 1) | (me ^ g[r, (c + 1) % X])
    return _pack(nib, X, Y)


def site_nibbles(arr_black_sites, arr_white_sites):
    """(Y, X) matrix of the nibble each lattice site's update applies, from the packed arrays holding the black / white sites' nibbles."""
    Y, lld = arr_black_sites.shape
    X = lld * 32
    out = np.zeros((Y, X), dtype=np.uint8)
    sh = (np.arange(16, dtype=np.uint64) * np.uint64(4))[None, None, :]
    for color, arr in ((0, arr_black_sites), (1, arr_white_sites)):
        nib = ((arr[:, :, None] >> sh) & np.uint64(0xF)).astype(np.uint8).reshape(Y, X // 2)
        out[np.arange(Y)[:, None], _cols(X, Y, color)] = nib
    return out


def bonds_symmetric(N):
    """J_ij == J_ji for every bond of the periodic lattice: a site's <up> bit is its upper neighbour's <down> bit, its <left> bit
    its left neighbour's <right> bit."""
    up, down, left, right = (N >> 3) & 1, (N >> 2) & 1, (N >> 1) & 1, N & 1
    return bool(np.array_equal(up, np.roll(down, 1, axis=0)) and np.array_equal(left, np.roll(right, 1, axis=1)))
