"""A SECOND, independent restatement of the reference's -J coupling generators, in numpy, written from the
closed form instead of the kernels' control flow -- test infrastructure like oracle/, used by tests/ only:

  ham_black_np   hamiltInitB_k (optimized/main.cu:153-212, launched :1729-1736 with seed + 1).  oracle/ising_oracle.c walks the
                 reference's thread blocks and pulls draws from a sequential generator; here every coupling BIT is addressed
                 directly: the bit (row, packed word w, nibble k, direction l) is draw number d of Philox subsequence tid,
                     v = w // 2 (128-bit vector), xy = w % 2, bx = v // 32, tx = v % 16, j = (v % 32) // 16   (LOOP_X = 2, :163-169)
                     tid = ((row // 16) * gx + bx) * 256 + (row % 16) * 16 + tx                                  (:171-172, global row)
                     d   = ((j * 16 + k) * 4 + l) * 2 + xy          (order j, nibble k, bit l, word x then y: :186-200)
                 i.e. output d % 4 of the Philox4x32-10 block with counter (d // 4, 0, tid, 0) and key = seed (curand_init(seed, tid,
                 0), :175), set where curand_uniform < prob (:193, :196).
  ham_white_np   hamiltInitW_k (:214-331).  The kernel scatters masks of the black words with shifts and atomicOr; here the
                 statement is the physical one it implements: a bond has ONE coupling bit, stored at its black end in the direction
                 of the white end -- the white end carries the same bit in the opposite direction.  Directions <right, left, down,
                 up> = bits 0..3 (:588-612).  Colour-site k of row r is lattice column 2k + (r & 1) black, 2k + 1 - (r & 1)
                 white; rows wrap every YSL rows (:306-307), columns every XSL columns (:320-321).
"""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)
LO = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Philox4x32-10 (Salmon et al., SC'11) on arrays of uint32 counters; returns the four output arrays."""
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint32) for c in (c0, c1, c2, c3))
    k0, k1 = np.uint32(k0), np.uint32(k1)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = M0 * c0.astype(np.uint64)
            p1 = M1 * c2.astype(np.uint64)
            n0 = (p1 >> np.uint64(32)).astype(np.uint32) ^ c1 ^ k0
            n1 = (p1 & LO).astype(np.uint32)
            n2 = (p0 >> np.uint64(32)).astype(np.uint32) ^ c3 ^ k1
            n3 = (p0 & LO).astype(np.uint32)
            c0, c1, c2, c3 = n0, n1, n2, n3
            k0, k1 = np.uint32(k0 + W0), np.uint32(k1 + W1)
    return c0, c1, c2, c3


def curand_uniform(x):
    """x * 2^-32 + 2^-33 in FP32: the product is exact, one rounding in the sum (cuRAND's _curand_uniform)."""
    return (x.astype(np.float32) * np.float32(2.0 ** -32) + np.float32(2.0 ** -33)).astype(np.float32)


def ham_black_np(X, Y, row_base, seed, prob):
    """Rows [row_base, row_base + Y) of the black coupling array of a lattice X columns wide, as [Y][X/32] uint64 (4 bits per site).
    `seed` is the generator's seed, i.e. the run's seed + 1 (:1734)."""
    gx, lld = X // 2048, X // 32
    row = (row_base + np.arange(Y, dtype=np.int64))[:, None, None, None]           # global row
    w = np.arange(lld, dtype=np.int64)[None, :, None, None]
    k = np.arange(16, dtype=np.int64)[None, None, :, None]
    l = np.arange(4, dtype=np.int64)[None, None, None, :]
    v, xy = w // 2, w % 2
    bx, tx, j = v // 32, v % 16, (v % 32) // 16
    tid = ((row // 16) * gx + bx) * 256 + (row % 16) * 16 + tx
    d = ((j * 16 + k) * 4 + l) * 2 + xy
    tid, d = np.broadcast_arrays(tid, d)
    blk = d // 4
    out = philox4x32_10(blk & 0xFFFFFFFF, blk >> 32, tid & 0xFFFFFFFF, tid >> 32, seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    x = np.choose(d % 4, out)
    bit = (curand_uniform(x) < np.float32(prob)).astype(np.uint64)
    sh = (4 * k + l).astype(np.uint64)
    return (bit << np.broadcast_to(sh, bit.shape)).sum(axis=(2, 3), dtype=np.uint64)


def _unpack(words):
    """[Y][X/32] uint64 -> [Y][X/2] nibbles (one per colour-site)."""
    Y, lld = words.shape
    sh = (np.arange(16, dtype=np.uint64) * np.uint64(4))[None, None, :]
    return ((words[:, :, None] >> sh) & np.uint64(0xF)).astype(np.uint8).reshape(Y, lld * 16)


def _pack(nib):
    Y, n = nib.shape
    sh = (np.arange(16, dtype=np.uint64) * np.uint64(4))[None, None, :]
    return (nib.astype(np.uint64).reshape(Y, n // 16, 16) << sh).sum(axis=2, dtype=np.uint64)


RIGHT, LEFT, DOWN, UP = 1, 2, 4, 8


def ham_white_np(hamB, XSL=0, YSL=0):
    """The white coupling array of the WHOLE lattice from its black one: every white site collects, from its four black neighbours,
    the bit each of them stores for the bond between the two, in the opposite direction.  Periodic per XSL x YSL block."""
    B = _unpack(hamB)                      # B[r, k]: nibble of black colour-site k of row r
    Y, n = B.shape                         # n = X / 2 colour-sites per row
    ysl = YSL or Y
    xs = (XSL or 2 * n) // 2               # colour-sites per sub-lattice row
    r = np.arange(Y)[:, None]
    k = np.arange(n)[None, :]
    up_r = np.where(r % ysl == 0, r + ysl - 1, r - 1)
    dn_r = np.where((r + 1) % ysl == 0, r - ysl + 1, r + 1)
    # a white site's vertical neighbours are the black sites with the SAME colour index in the rows above / below
    W = np.where(B[up_r, k] & DOWN, UP, 0) | np.where(B[dn_r, k] & UP, DOWN, 0)
    # horizontal: in an even row the lattice reads b0 w0 b1 w1 ..., in an odd row w0 b0 w1 b1 ...  (periodic every xs colour-sites)
    k_next = np.where((k + 1) % xs == 0, k + 1 - xs, k + 1)
    k_prev = np.where(k % xs == 0, k + xs - 1, k - 1)
    even = (r % 2 == 0)
    left_black = np.where(even, k, k_prev)     # colour index of the black site to the LEFT of white site k
    right_black = np.where(even, k_next, k)    # ... to its RIGHT
    rr = np.broadcast_to(r, left_black.shape)
    W = W | np.where(B[rr, left_black] & RIGHT, LEFT, 0) | np.where(B[rr, right_black] & LEFT, RIGHT, 0)
    return _pack(W.astype(np.uint8))
