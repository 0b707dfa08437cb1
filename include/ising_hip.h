/*
 * ising_hip.h -- C-ABI of libising_hip.so: the MI355X (gfx950) checkerboard-Metropolis engine.
 *
 * The reference (NVIDIA/ising-gpu, optimized/main.cu) has no library or FFI surface: its boundary is the
 * process (argv in, transcript out) and, inside main(), the kernel-launch sites.  This header is the drop-in
 * boundary for those launch sites: every entry point names the reference code it replaces (file:line are
 * relative to /root/reference).  Plain pointers and sizes only; no torch / C++ types.
 *
 * Conventions
 *   - All functions return 0 on success and a non-zero ISING_E_* code on failure; ising_last_error() gives a
 *     human-readable message for the calling thread's last failure.  (The reference prints and calls
 *     exit(EXIT_FAILURE), optimized/cudamacro.h:25-39; the CLI front does that with these codes.)
 *   - A context owns one *slab*: rows [slab*Y, (slab+1)*Y) of both colour arrays of a lattice that is
 *     nslabs*Y rows by X columns (optimized/main.cu:1627-1628).  nslabs == 1 is the single-GPU case.
 *   - Packed layout is the reference's: per colour, row-major [Y][X/32] 64-bit words, 16 spins per word,
 *     4 bits per spin, bit 0 of each nibble = spin (1 = up) (optimized/main.cu:40, :1243, :1539).
 *   - All device work is enqueued on the context's stream (default: the legacy default stream, 0) and is
 *     asynchronous unless stated; the context is not re-entrant (one host thread per context, as the
 *     reference drives all devices from one thread, optimized/main.cu:1764-1805).
 */
#ifndef ISING_HIP_H
#define ISING_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ISING_BLACK 0  /* enum {C_BLACK, C_WHITE}, optimized/main.cu:80 */
#define ISING_WHITE 1
#define ISING_HAM_BLACK 2 /* the black coupling array (-J), as a third "colour" for ising_halo_ptrs / ising_ring_exchange */

#define ISING_CRIT_TEMP 2.26918531421f /* CRIT_TEMP, optimized/main.cu:42 */
#define ISING_SEED_DEF 463463564571ull /* SEED_DEF, optimized/main.cu:63 */

enum {
	ISING_OK = 0,
	ISING_E_ARG = 1,      /* bad argument (sizes not multiples of 2048 / 16, null pointer, ...) */
	ISING_E_HIP = 2,      /* a HIP runtime call failed */
	ISING_E_STATE = 3,    /* call sequence error (e.g. sweep with nslabs > 1); also: a fused launch gave up waiting for completion
	                         counters that never came (reported by the next call that synchronises; tickets and counters are reset) */
	ISING_E_NOGPU = 4,    /* no usable gfx950 device / kernel image */
	ISING_E_RCCL = 5,     /* RCCL could not be opened or an RCCL call failed (ring transports) */
	ISING_E_TIMEOUT = 6,  /* ising_rank_wait: the exchange did not complete in time */
	ISING_E_IO = 7        /* checkpoint / dump file error */
};

/* kernel selection for ising_update_color / ising_sweep (A/B and fallback); all variants give identical results */
enum {
	ISING_KERNEL_AUTO = 0,    /* integer-threshold kernel (v_cmpx form) when the temperature admits it, else generic */
	ISING_KERNEL_GENERIC = 1, /* per-site FP32 compare against the exp table, exactly as the reference writes it */
	ISING_KERNEL_FAST = 2     /* integer thresholds compared per site (v_cmpx); error if thresholds do not fit */
};

/* device layout of the spin arrays.  The C-ABI always speaks the reference's packed layout (read/write/dump convert). */
enum {
	ISING_LAYOUT_AUTO = 0,   /* ballot where it applies and pays (from 1.5 * 2^24 spins per slab where fused launches apply: a slab that wraps
	                            in place -- that one also below, while a level of one-row units still feeds two workgroups per CU, e.g.
	                            8192 x 1280, 16384 x 768 --, a ring slab that can keep ghost rows; from 2^27 otherwise), else dense */
	ISING_LAYOUT_NIBBLE = 1, /* the reference's: 4 bits per spin, 16 spins per 64-bit word (optimized/main.cu:40, :1243) */
	ISING_LAYOUT_DENSE = 2,  /* 1 bit per spin, 32 spins per 32-bit word = one reference 128-bit vector per word */
	ISING_LAYOUT_BALLOT = 3  /* 1 bit per spin, 64-bit words in the update kernel's wave-ballot order (ising_ballot.hip),
	                            rows padded to whole wave columns of 8192 lattice columns; with sub-lattices, -J or a
	                            caller-owned buffer X % 8192 == 0 is required (sub-lattice widths of 2048, 4096 or a
	                            multiple of 8192); a temperature without integer accept thresholds turns the slab into
	                            ISING_LAYOUT_DENSE at the next update (pointers from ising_halo_ptrs / ising_device_ptr
	                            are void after that when the rows were padded) */
};

typedef struct ising_ctx ising_ctx;

typedef struct ising_config {
	int32_t X;        /* lattice columns (black+white spins); multiple of 2048 (optimized/main.cu:1314, :1412-1416) */
	int32_t Y;        /* rows of THIS slab; multiple of 16 (optimized/main.cu:1317, :1417-1421) */
	int32_t nslabs;   /* slabs in the periodic ring = the reference's ndev (-d, optimized/main.cu:1335) */
	int32_t slab;     /* index of this slab = the reference's devid kernel argument */
	uint64_t seed;    /* -s (optimized/main.cu:1329) */
	float temp;       /* absolute temperature, float as in the reference (optimized/main.cu:1342, :1465-1471) */
	int32_t device;   /* HIP device ordinal that holds this slab */
	int32_t strip_rows; /* rows each thread marches per launch; 0 = choose automatically */
	int32_t kernel;   /* ISING_KERNEL_* */
	int32_t XSL;      /* --xsl: sub-lattice columns (multiple of 2048 dividing X); 0 = no sub-lattices */
	int32_t YSL;      /* --ysl: sub-lattice rows (multiple of 16 dividing Y).  With sub-lattices every XSL x YSL block of
	                     the slab is an independent periodic system (optimized/main.cu:1423-1462; loadTile wrap arguments
	                     slX, slY :413-459), so no halo exchange is needed. */
	void *lattice_mem;   /* optional caller-owned device buffer of ising_required_bytes() bytes for the spin arrays (e.g. a
	                        torch tensor, so that its edge/halo rows can be handed to RCCL as ordinary tensors); NULL = the
	                        library allocates.  (A lone slab of up to 2^24 spins on ISING_LAYOUT_DENSE sweeps in tile launches that
	                        alternate between this buffer and a library-owned twin of the same size, allocated by ising_create; the
	                        spins are back in this buffer whenever a call returns.  The quad path -- lone lattices of one to eight
	                        blocks of 2048 columns -- keeps four colour arrays of its own, 4 x X*Y/16 bytes, plus the accept masks of two
	                        passes, 2 x 2T levels x X*Y/16 bytes at T sweeps a pass (T <= 16: up to 4.5 x X*Y bytes in all), next to it) */
	void *coupling_mem;  /* the same for the -J coupling arrays */
	int32_t layout;   /* ISING_LAYOUT_* */
	int32_t use_J;    /* -J given: allocate coupling arrays and apply them in every update (useGenHamilt, :1368-1372) */
	float J_prob;     /* -J <PROB>: probability that a bond is anti-ferromagnetic, clamped to [0,1] (:1370) */
	int32_t ring_halo; /* nslabs == 1 only: 1 = treat the slab as a ring of one, i.e. rows -1 / Y are halo rows that the ring
	                      transport fills with the slab's own last / first row instead of mirrors the kernels maintain (runs
	                      the transports, RCCL included, on a single GPU); ignored for nslabs > 1 */
	size_t lattice_mem_bytes;  /* size of lattice_mem / coupling_mem when given (checked against ising_required_bytes_layout; */
	size_t coupling_mem_bytes; /* 0 = trust the caller, as the reference trusts cudaMalloc) */
} ising_config;

const char *ising_last_error(void);
/* Every ISING_* environment switch the library (and its Python mirror) reads, as a markdown table: name, values, meaning and default.  Switches are A/B and test
 * aids, read once per context in ising_create; docs/SWITCHES.md is this table, tests/test_switches.py holds both against the sources.  buf may be NULL: *needed
 * = bytes including the terminator; ISING_E_ARG when len is too small (the text is truncated, *needed says how much it takes).  Needs no GPU. */
int ising_switch_table(char *buf, size_t len, size_t *needed);

/* Number of visible HIP devices (cudaGetDeviceCount at optimized/main.cu:1481-1491). */
int ising_device_count(int *count);
/* Device description for the "Using GPUs" block (optimized/main.cu:1482-1490). */
int ising_device_info(int device, char *name, size_t name_len, int *cus, int *max_threads_per_cu, int *major, int *minor);
/* One cell of the "GPUs direct access matrix" (cudaDeviceCanAccessPeer, optimized/main.cu:1508-1537): *can_access = 1 when `device` can load and store
 * `peer`'s memory directly (a device reaches itself: 1).  Asks only; the ring enables peer access where it uses it. */
int ising_device_peer_access(int device, int peer, int *can_access);

/* Measurement aid for the bench's roofline object: sites per nanosecond of a kernel that only DRAWS -- one Philox4x32-10
 * output per site exactly as the update kernels generate them, no accept test, no lattice, no memory traffic.  The
 * update kernels are bound by the vector ALU, so this is their ceiling at bit-exact parity.  Blocking (~30 ms). */
int ising_philox_ceiling(int device, double *sites_per_ns);
/* The same as an AVERAGE over launches that last min_ms together (the update kernel's figure is an average over a long launch too), with the
 * shader clock the launches ran at: *sclk_mhz = cycles of the shader's counter / time of the constant 100 MHz counter, both read inside the
 * kernel by one wave per XCD (0 when the counters make no sense).  Blocking. */
int ising_philox_ceiling_clocked(int device, double min_ms, double *sites_per_ns, double *sclk_mhz);
/* Measurement aid: enable != 0 makes the first eight workgroups (one per XCD) of every fused launch of this slab leave the same two counters when
 * they start and when they leave; ising_kernel_clock_fetch (blocking) turns the LAST launch's marks into the clock it ran at -- mean / min / max over
 * the XCDs (any pointer may be NULL).  ISING_E_STATE when no fused launch has run since it was enabled.  bench.py reports it next to the ceiling's. */
int ising_kernel_clock(ising_ctx *ctx, int enable);
int ising_kernel_clock_fetch(ising_ctx *ctx, double *sclk_mhz_mean, double *sclk_mhz_min, double *sclk_mhz_max);

/* Bytes of one caller-supplied device buffer.  ising_required_bytes: 2 colours x (Y + 2) rows x X/4 bytes -- the
 * reference's 4 bit/spin, enough for every layout and the size of the coupling buffer (coupling_mem).
 * ising_required_bytes_layout: what the spin arrays of a given device layout occupy (ISING_LAYOUT_NIBBLE: the same;
 * ISING_LAYOUT_DENSE / _BALLOT / _AUTO: a quarter -- AUTO only ever picks one of the two 1 bit/spin layouts). */
size_t ising_required_bytes(int32_t X, int32_t Y);
size_t ising_required_bytes_layout(int32_t X, int32_t Y, int32_t layout);

/* Allocates the slab (both colours, zeroed), halo-receive rows and the threshold/exp tables.
 * Replaces the cudaMalloc/cudaMallocManaged + memset + exp_d upload of optimized/main.cu:1599-1703. */
int ising_create(const ising_config *cfg, ising_ctx **out);
/* Frees everything the context owns (optimized/main.cu:1900-1924). */
int ising_destroy(ising_ctx *ctx);

/* Use an externally created hipStream_t (e.g. torch's current stream) for all subsequent work.  Call it on an idle
 * context (ising_synchronize first): work already enqueued is not re-ordered behind the new stream. */
int ising_set_stream(ising_ctx *ctx, void *hip_stream);
/* Gives the context a non-blocking stream of its own (created here, destroyed with the context).  Contexts share the
 * device's default stream otherwise and their launches serialise; independent lattices on one GPU (replicas at other
 * temperatures: cuIsing --tsweep) run side by side on private streams -- a lattice of 8192^2 fills 70 % of the chip, two
 * or three of them 90 %.  (New: the reference runs one lattice per process, optimized/main.cu:1596-1598.) */
int ising_use_private_stream(ising_ctx *ctx);
/* Blocks until all work enqueued by this context has finished (cudaDeviceSynchronize, optimized/main.cu:1751-1754). */
int ising_synchronize(ising_ctx *ctx);

/* latticeInit_k<BLACK> + latticeInit_k<WHITE> for this slab (optimized/main.cu:92-151, launches :1708-1726). */
int ising_init_lattice(ising_ctx *ctx);

/* -J: hamiltInitB_k (seed+1) for this slab, then hamiltInitW_k (optimized/main.cu:153-331, launches :1729-1742).
 * ising_init_couplings does both and needs nslabs == 1.  With several slabs the white array needs the neighbours'
 * black edge rows: call ..._black on every slab, deliver the ISING_HAM_BLACK halo rows (ising_halo_ptrs /
 * ising_ring_exchange), then ..._white.  ising_ring_init_couplings does all of that for a single-process ring. */
int ising_init_couplings(ising_ctx *ctx);
int ising_init_couplings_black(ising_ctx *ctx);
int ising_init_couplings_white(ising_ctx *ctx);
/* Copies rows of a coupling array (which = ISING_BLACK / ISING_WHITE for hamB / hamW) to host memory. Blocking. */
int ising_read_couplings(ising_ctx *ctx, int which, int64_t row0, int64_t nrows, uint64_t *dst_host);
/* The other way round (new: the reference only draws its couplings at random): a whole coupling array of a lattice that wraps in
 * place (nslabs == 1), Y rows of X/32 words in the reference's form -- one nibble per site, bits <up, down, left, right>
 * (optimized/main.cu:588-612: a set bit flips that neighbour's spin before the energy sum = an antiferromagnetic bond) -- is
 * copied to the device and brought into the form the context's update kernels read.  Which array an update reads is the
 * reference's choice: the BLACK update takes its sites' nibbles from array ISING_WHITE (hamW) and the WHITE update from array
 * ISING_BLACK (hamB), each at the destination's own packed index (the launches at :1774 and :1795).  For a model with
 * symmetric bonds J_ij = J_ji write the black sites' bonds into ISING_WHITE and the white sites' into ISING_BLACK; the arrays
 * ising_init_couplings draws are the other way round, as in the reference (hamB = the black sites' bonds, hamW = the white
 * sites' bonds gathered from them, :214-331): with them a site applies the bonds of its horizontal neighbour.  Blocking. */
int ising_write_couplings(ising_ctx *ctx, int which, const uint64_t *src_host);
/* The two coupling arrays of a slab change places (their halo / ghost rows included): after ising_init_couplings /
 * ising_ring_init_couplings every colour's update then reads its OWN sites' bonds -- the Edwards-Anderson +-J model with
 * symmetric bonds -- instead of its horizontal neighbour's, which is what the reference's launches give (above).  Results then
 * differ from the reference's, by design (cuIsing --J-symmetric).  Call it on every slab of a ring. */
int ising_swap_couplings(ising_ctx *ctx);

/* Recomputes the exp table / integer thresholds (optimized/main.cu:1684-1703, temperature ramp :1848-1859). */
int ising_set_temperature(ising_ctx *ctx, float temp);
/* The ten FP32 table entries exp_h[2][5] in use (optimized/main.cu:1681-1697) and the derived integer
 * thresholds: thr[a] = number of 32-bit draws x with curand_uniform(x) <= table value for `a` aligned
 * neighbours (0..4); 2^32 means "always flips". */
int ising_get_tables(ising_ctx *ctx, float exp_table[10], uint64_t thr[5]);

/* One colour half-sweep, spinUpdateV_2D_k<COLOR> (optimized/main.cu:463-670; launches :1766-1777, :1787-1798)
 * restricted to rows [row_lo, row_hi) of this slab.  `it` is the reference's 1-based iteration argument (j+1).
 * Rows 0 and Y-1 read the halo rows of the opposite colour: for nslabs == 1 these mirror the slab itself (periodic
 * wrap), otherwise they must have been delivered into the buffers returned by ising_halo_ptrs before the launch
 * runs.  Every row is independent of the others within a colour, so any partition gives the same result. */
int ising_update_color(ising_ctx *ctx, int it, int color, int row_lo, int row_hi);
/* The same for just the two edge rows 0 and Y-1 in one launch: what the neighbours need first, so that the halo
 * exchange can overlap with ising_update_color(ctx, it, color, 1, Y-1). */
int ising_update_edges(ising_ctx *ctx, int it, int color);
/* Rows each lane marches per launch (cfg.strip_rows or the automatic choice) and the resulting strip count. */
int ising_strip_info(ising_ctx *ctx, int *strip_rows, int *nstrips);

/* nslabs == 1 only: `nsweeps` full sweeps, black then white, iterations first_it .. first_it+nsweeps-1
 * (the hot loop, optimized/main.cu:1763-1805).  How the sweeps are launched depends on the lattice (ising_sweep_info): fused
 * launches of many sweeps on the ballot layout (from 1.5 * 2^24 spins, and below for lattices of enough rows), one launch per pass of 4 - 8 sweeps on the quad
 * layout for lone lattices of one to eight blocks of 2048 columns (round 5: a word pass on tiles + halo next to the draws of the pass to come; the spins are
 * converted from and to the dense layout at either end of the call), tile launches of 3 - 6
 * sweeps on the dense layout up to 2^24 spins (ising_create allocates a second lattice buffer of the size of the first -- a sweep never allocates; without it
 * the call returns ISING_E_STATE --: every launch reads one and writes the other, and an even number of launches per call leaves the spins where every other entry
 * point expects them), one launch per
 * colour otherwise -- the spins after the call are the same in every form. */
int ising_sweep(ising_ctx *ctx, int first_it, int nsweeps);
/* The hot loop WITH its print points (optimized/main.cu:1763-1810: the sweeps, and countSpins whenever the iteration is a multiple of
 * printFreq -- every number the reference publishes was measured with `-p 16` inside the timed loop).  nslabs == 1 only: `nsweeps` full
 * sweeps, iterations first_it .. first_it+nsweeps-1; ups[k] = the number of up spins after the k-th iteration `it` of them with
 * it % every == 0 (at most max_counts; *ncounts = how many).  Where ising_sweep issues fused launches the counts are taken INSIDE the
 * launches, by the units that store the words -- no launch boundary, no count kernel and no read-back between two print points (16384^2
 * with a count every 16 sweeps: 3057 -> 3290 flips/ns) --; on the other layouts and with sub-lattices or couplings the call sweeps and
 * counts in turn.  Blocking: the counts are read back when the last launch is done.
 * bond_equal (NULL: no energy): bond_equal[k] = ising_bond_equal's sum A at the same print point -- north_star's energy series next to the
 * magnetisation's, e = -(2A - 2N)/N; the reference computes no energy.  Inside the fused launches too: the white level of a measured
 * sweep counts, per stored word, the neighbours that equal it (four XOR + popcount per row on one level in 2 * every: `cuIsing -p 16 --energy`
 * costs what `-p 16` costs). */
int ising_sweep_counted(ising_ctx *ctx, int first_it, int nsweeps, int every, uint64_t *ups, int64_t *bond_equal, int max_counts, int *ncounts);
/* How ising_sweep launches right now: *fused = 1 when it issues fused launches (ballot layout: from 1.5 * 2^24 spins up and where ISING_LAYOUT_AUTO picks it below, or
 * ISING_FUSED=1: one launch carries up to *max_sweeps_per_launch sweeps = twice as many colour half-sweeps, handed out to
 * a chip-filling grid through in-order tickets; ising_ballot.hip), 0 when it issues one launch per colour, 2 when it issues tile
 * launches (a lone slab on the dense layout up to 2^24 spins, or ISING_TILES=1: every workgroup sweeps a tile + halo of its own
 * *max_sweeps_per_launch times without a word from the others; ising_dense.hip: dense_tile_k).  For a ring slab with ghost rows G
 * deep (ballot layout): how the ring sweeps it -- fused launches of up to G/2 sweeps between exchanges.  3: fused launches, and calls of 2^35 flips and more
 * (ISING_SPLIT=1: every call) run them in the split form -- a call shorter than that still issues the plain fused form at its own strip height; ising_sweep_form
 * answers for a given call --
 * (round 5; ising_ballot.hip: ballot_split_k): draw units -- no wait, no barrier, tall strips -- and word units with tickets of their own, for lone
 * lattices and ring slabs whose levels have too few tickets for tall strips in the plain fused form.  4: the quad path (round 5; ising_quad.hip: quad_pass_k) --
 * one launch per pass of *max_sweeps_per_launch sweeps: tiles of a few row groups x the whole width + halo in LDS run the pass's word phases while the rest of
 * the chip draws the accept masks of the pass to come, once (lone lattices of one to eight blocks of 2048 columns and few enough rows, or ISING_QUAD=1). */
int ising_sweep_info(ising_ctx *ctx, int *fused, int *max_sweeps_per_launch);
/* The launch form and shape ising_sweep(ctx, ., nsweeps) issues for a call of that many sweeps: *form as ising_sweep_info's *fused, except that 3 is answered only
 * when THIS call runs split launches (1 otherwise); *strip_rows, *wg_per_cu: strip height and workgroups per CU of the launches of that call (0 where the form has
 * none: one launch per colour, tiles, quad).  Any pointer may be NULL. */
int ising_sweep_form(ising_ctx *ctx, int nsweeps, int *form, int *strip_rows, int *wg_per_cu);
/* The run-time guard under the fused launches' shape table (round 6).  Strip height and workgroups per CU of a lone slab's fused launches come from tables fitted
 * on the boxes this library was measured on, and some mid-size entries sit next to cliffs (the same shape: 3392 flips/ns on one box, 835 on the next).  The first
 * launches of ising_sweep on such a slab are therefore timed on their dispatch packets -- the host waits for them, once per context.  On the plateau (4096 tickets
 * a level and more: 32768^2 and up) a rate of 0.8 x what a lattice of that size runs at settles it after one launch; below, or under that rate, the neighbouring
 * shapes (a workgroup per CU fewer, one more -- on down the slope while it pays --, half the strip height) get one launch each and the fastest stays if it is
 * worth 3 %.  Where the table gives long calls the split form, one split and one fused launch are timed the same way and the faster form stays (round 5's boxes:
 * split 1-6 % ahead; one of round 6's: 3-5 % behind at 65536 x 8192 and 24576^2).  Results never depend on the shape or the form.  ISING_GUARD=0 turns it off (default: on, on a whole MI355X); ising_sweep_form reports the shape in
 * force.  state: 0 off (not a lone slab in the fused form), 1 timing the table's shape, 2 trying neighbours, 3 settled.  Blocks while a timed launch is in flight. */
typedef struct ising_guard_info {
	int32_t state, switched, launches_timed;
	int32_t table_strip_rows, table_wg_per_cu;  /* the shape ising_create (or ISING_FUSED_WGS / strip_rows) gave */
	int32_t strip_rows, wg_per_cu;              /* the shape in force */
	float expected_flips_per_ns;                /* what the first launches were held against (x 0.8) */
	float table_flips_per_ns, kept_flips_per_ns; /* measured: the table's shape, the shape that stayed */
	/* the FORM of a long call's launches where the table says "split" (ising_sweep_info: 3): one launch of each form is timed, the faster stays (2 %) */
	int32_t form_state;                         /* -1: no split form here (or ISING_SPLIT=1: always), 0: nothing timed yet, 1: a split launch timed, 2: a fused one in flight, 3: decided */
	int32_t split_kept;                         /* long calls run split launches */
	float split_flips_per_ns, fused_flips_per_ns;
} ising_guard_info;
int ising_shape_guard_info(ising_ctx *ctx, ising_guard_info *out);
/* Same, bracketed by HIP events on the context's stream; returns elapsed milliseconds (blocking). */
int ising_sweep_timed(ising_ctx *ctx, int first_it, int nsweeps, float *elapsed_ms);

/* Halo exchange surface for nslabs > 1 (replaces the managed-memory remote loads of
 * optimized/main.cu:1637-1642 / loadTile :413-428).  For colour `color`:
 *   send_top / send_bot : device pointers to this slab's first / last row (row_bytes each), to be sent to the
 *                         previous / next slab in the ring;
 *   recv_top / recv_bot : device buffers that must receive the previous slab's last row / the next slab's
 *                         first row before a half-sweep of the OTHER colour touches rows 0 / Y-1. */
int ising_halo_ptrs(ising_ctx *ctx, int color, void **send_top, void **send_bot, void **recv_top, void **recv_bot,
                    size_t *row_bytes);
/* The same surface for the DEEP exchange the library's own ring uses (ising_ring_sweep / ising_rank_sweep), for callers
 * that bring their own transport (MPI, torch.distributed, ...).  A ring slab on the ballot layout that owns its buffer
 * keeps *depth = G ghost rows on either side (64, or Y/2 on flat slabs; 1 = no such rows: use the one-row surface above):
 *   send_top / send_bot : this slab's first / last G rows of `color` (block_bytes each, contiguous) -> previous / next slab;
 *   recv_top / recv_bot : the ghost rows above row 0 / below row Y-1 <- the previous slab's last / the next slab's first G rows.
 * Once both blocks of a colour have been filled -- by work ordered before the context's stream --, say so with
 * ising_ghost_delivered.  With both colours delivered, ising_sweep_ghost runs `nsweeps` <= G/2 full sweeps as ONE fused
 * launch over the slab and its ghost rows (whose draws are the ones their owners make: the generator is counter-based), so
 * that the ring costs one exchange of G rows per colour every G/2 sweeps instead of one row per colour half-sweep
 * (optimized/main.cu:1779-1805 synchronises all devices after every colour).  Whatever changes the spins (a sweep, an
 * init, a write) makes the ghost rows stale: ising_sweep_ghost then fails with ISING_E_STATE until they are delivered again. */
int ising_ghost_ptrs(ising_ctx *ctx, int color, int *depth, void **send_top, void **send_bot, void **recv_top, void **recv_bot,
                     size_t *block_bytes);
int ising_ghost_delivered(ising_ctx *ctx, int color);
int ising_sweep_ghost(ising_ctx *ctx, int first_it, int nsweeps);

/* countSpins / getMagn_k for this slab (optimized/main.cu:701-734, :831-868): number of up and down spins.
 * Blocking (copies two 64-bit counters back, as :860-866 does). */
int ising_count(ising_ctx *ctx, uint64_t *up, uint64_t *down);

/* Build-side observable (the reference never computes energy): A = sum over this slab's black sites of the
 * number of white neighbours equal to the site.  Over all slabs, sum_<ij> s_i s_j = 2A - 2N.  Reads the white
 * halo rows like ising_update_color(BLACK).  Blocking. */
int ising_bond_equal(ising_ctx *ctx, int64_t *A);

/* Copies rows [row0,row0+nrows) of one colour to / from host memory in the packed layout
 * (the D2H copy of dumpLattice, optimized/main.cu:1150-1152).  Blocking. */
int ising_read_packed(ising_ctx *ctx, int color, int64_t row0, int64_t nrows, uint64_t *dst_host);
int ising_write_packed(ising_ctx *ctx, int color, int64_t row0, int64_t nrows, const uint64_t *src_host);

/* The same rows at 1 bit per spin: X/64 32-bit words per row, one word per reference 128-bit vector (ulonglong2, :1716):
 * bit k = nibble k of its word x (k < 16) / nibble k-16 of its word y.  A quarter of the packed size. */
int ising_read_bits(ising_ctx *ctx, int color, int64_t row0, int64_t nrows, uint32_t *dst_host);
int ising_write_bits(ising_ctx *ctx, int color, int64_t row0, int64_t nrows, const uint32_t *src_host);

/* Device pointer to row 0 of a colour array of this slab in its DEVICE layout (see ising_layout) for zero-copy
 * consumers: nibble layout = [Y][X/32] 64-bit words exactly as the reference's buffers; dense = [Y][X/64] 32-bit words;
 * ballot = [Y][64 * ceil(X/8192)] 64-bit words in the bit order described in ising_ballot.hip. */
int ising_device_ptr(ising_ctx *ctx, int color, void **ptr, size_t *bytes);
/* Asynchronous measurement: enqueues the up count (ising_count) and the bond sum (ising_bond_equal) of the state the
 * context's stream holds at this point; nothing waits.  ising_measure_fetch waits for the stream and hands back all pending
 * results in enqueue order (at most 4096 may be pending).  A (sweeps, measurement) series then runs without a host round
 * trip in between -- the reference reads its counters back at every print point (optimized/main.cu:1806-1810). */
int ising_measure_enqueue(ising_ctx *ctx);
int ising_measure_fetch(ising_ctx *ctx, uint64_t *up, int64_t *bond_equal, int max_n, int *n);

/* The layout in use right now (ISING_LAYOUT_NIBBLE, _DENSE or _BALLOT). */
int ising_layout(ising_ctx *ctx, int *layout);

/* dumpLattice (optimized/main.cu:1140-1209): writes "<prefix><slab>.txt", one text row per lattice row, one
 * hex digit per spin, colours interleaved by row parity. */
int ising_dump_text(ising_ctx *ctx, const char *prefix);

/* ---- binary checkpoint (the reference has none: its only dump is the text file above).  One file for the
 * whole lattice: a 136-byte header (X, total rows, seed, completed sweeps `it`, temperature, sub-lattice / coupling
 * settings), all black rows then all white rows in GLOBAL row order at 1 bit per spin (the ising_read_bits format), and
 * the number of up spins as a check -- so a file written by 8 slabs can be loaded into 1, 2, 4 ... slabs of the same
 * lattice.  Couplings (-J) are not stored: they are regenerated from the seed.  ctxs[k] = slab k of n (n = 1: &ctx).
 * save: blocking, writes "<path>.part" and renames.  load: checks geometry, seed and settings, fills every slab, verifies
 * the up-spin count on the device; the caller then delivers the halo rows (ising_ring_exchange, both colours) and
 * continues with first_it = *it + 1. */
typedef struct ising_checkpoint_info {
	int32_t X, Y_total, nslabs_written, XSL, YSL, use_J;
	float temp, J_prob;
	uint64_t seed;
	int64_t it;
} ising_checkpoint_info;
int ising_checkpoint_info_read(const char *path, ising_checkpoint_info *info);
int ising_ring_checkpoint_save(ising_ctx **ctxs, int n, const char *path, int64_t it);
int ising_ring_checkpoint_load(ising_ctx **ctxs, int n, const char *path, int64_t *it);
/* The same file from / into a ring of processes (ising_rank_*): collective; every rank writes / reads its own rows at their
 * place in the global row order, so the file is the one a single process would write.  `path` must name the same file for every
 * rank.  After a load: ising_rank_exchange for both colours, continue with first_it = *it + 1.  The ranks agree on the outcome of
 * every stage: a call that fails on one rank returns an error on all of them (nobody is left waiting), and a file some rank could
 * not fill is removed instead of getting its name; after a failed load the ring's spins are undefined. */
int ising_rank_checkpoint_save(ising_ctx *ctx, const char *path, int64_t it);
int ising_rank_checkpoint_load(ising_ctx *ctx, const char *path, int64_t *it);

/* ---- the slab ring (replaces optimized/main.cu:1599-1658 managed memory + remote loads and the
 * cudaDeviceSynchronize barriers :1779-1784, :1800-1805).  Per colour half-sweep every slab updates its two edge rows
 * first, then its first / last row travel to the previous / next slab's halo rows on a second HIP stream per slab while
 * the interior rows are updated on the first; HIP events order the two streams (ising_ring.cpp).
 * Transports: RCCL (ncclSend / ncclRecv pairs inside one group; librccl is opened at run time) when every slab has its
 * own device, peer-to-peer device copies otherwise.  ISING_RING_TRANSPORT=copy|rccl|auto overrides AUTO. */
enum {
	ISING_TRANSPORT_AUTO = 0, /* RCCL when every slab has its own device and librccl opens, else copies */
	ISING_TRANSPORT_COPY = 1, /* hipMemcpyPeerAsync on the comm stream (single process only) */
	ISING_TRANSPORT_RCCL = 2, /* ncclSend / ncclRecv on the comm stream */
	ISING_TRANSPORT_IPC = 3   /* one process per slab, no RCCL: every rank copies its edge rows straight into the neighbours' halo / ghost
	                             rows, which it has mapped through hipIpcMemHandle (ising_ipc_export / ising_ipc_attach below) */
};

/* -- single process, n devices (the reference's process model: one host thread drives ndev GPUs, :1763-1805).
 * ctxs[k] must be slab k of nslabs == n contexts of one layout (any device placement). */
int ising_ring_set_transport(ising_ctx **ctxs, int n, int transport); /* optional; before the first exchange */
int ising_ring_transport(ising_ctx **ctxs, int n, int *transport);     /* the transport in use */
/* Delivers colour `color`'s first/last rows of every slab into the neighbours' halo rows (asynchronous): after
 * ising_init_lattice / ising_write_packed, for both colours, before the first sweep. */
int ising_ring_exchange(ising_ctx **ctxs, int n, int color);
/* -J for a whole ring: black couplings on every slab, their edge rows to the neighbours, then the white couplings. */
int ising_ring_init_couplings(ising_ctx **ctxs, int n);
/* `nsweeps` full sweeps over all slabs, iterations first_it .. first_it+nsweeps-1 (the hot loop).  Asynchronous. */
int ising_ring_sweep(ising_ctx **ctxs, int n, int first_it, int nsweeps);
/* The same WITH the reference's print points (optimized/main.cu:1806-1810: countSpins over all devices whenever the iteration is a multiple of
 * printFreq, inside the timed loop; see ising_sweep_counted): ups[k] = the up spins of the WHOLE lattice after the k-th iteration of the call
 * that is a multiple of `every`.  Where the ring sweeps its slabs through ghost rows with the exchanges in the launches' tails (ballot layout,
 * the default), every slab's launches count their own rows as they store them -- no launch boundary, count kernel or read-back between two
 * print points --; elsewhere the call sweeps and counts in turn.  bond_equal (NULL: no energy): the bond sum of the whole lattice at the same
 * points, as in ising_sweep_counted.  Blocking. */
int ising_ring_sweep_counted(ising_ctx **ctxs, int n, int first_it, int nsweeps, int every, uint64_t *ups, int64_t *bond_equal, int max_counts, int *ncounts);
/* Blocks until every slab's compute and comm streams are idle. */
int ising_ring_synchronize(ising_ctx **ctxs, int n);
/* Totals over the ring: countSpins (:831-868) and the bond sum of ising_bond_equal.  Blocking. */
int ising_ring_count(ising_ctx **ctxs, int n, uint64_t *up, uint64_t *down);
int ising_ring_bond_equal(ising_ctx **ctxs, int n, int64_t *A);

/* -- one process per GPU (torchrun / mpirun style launch): this process holds slab cfg.slab of cfg.nslabs.  Rank 0
 * creates an id, the launcher's own channel (torch.distributed, MPI, a file) carries its ISING_RCCL_ID_BYTES bytes to
 * every rank, every rank attaches (collective).  Afterwards the ising_rank_* calls are collective in the same sense as
 * the ring calls above: every rank makes the same calls in the same order. */
#define ISING_RCCL_ID_BYTES 128
int ising_rccl_available(int *version);            /* ISING_OK when librccl could be opened; its version code */
int ising_rccl_unique_id(void *id_out);            /* ncclGetUniqueId */
int ising_rank_attach(ising_ctx *ctx, const void *id_in); /* ncclCommInitRank(nslabs, id, slab) */
int ising_rank_detach(ising_ctx *ctx, int abort_pending);  /* ncclCommDestroy, or ncclCommAbort after a time-out (ISING_TRANSPORT_IPC:
                                                              unmaps the neighbours; abort_pending ends this rank's polling kernels) */
int ising_rank_exchange(ising_ctx *ctx, int color);
int ising_rank_init_couplings(ising_ctx *ctx);
int ising_rank_sweep(ising_ctx *ctx, int first_it, int nsweeps);
/* ising_ring_sweep_counted, one process per slab: collective (every rank calls it with the same arguments); the ranks' sums travel over the
 * rank transport (ncclAllReduce / the peer transport's shared-memory reduction). */
int ising_rank_sweep_counted(ising_ctx *ctx, int first_it, int nsweeps, int every, uint64_t *ups, int64_t *bond_equal, int max_counts, int *ncounts);
/* Waits until both streams of the slab are idle; timeout_ms >= 0 polls and returns ISING_E_TIMEOUT when the time is up
 * (a hung exchange can then be abandoned with ising_rank_detach(ctx, 1)); < 0 blocks. */
int ising_rank_wait(ising_ctx *ctx, int timeout_ms);
/* (Whatever makes a slab's ghost rows stale -- ising_init_lattice, ising_write_packed / _bits, ising_update_color -- is collective
 * in the same sense: ising_rank_sweep decides from the local state whether an exchange comes first, and ranks that disagree
 * wait for each other until ising_rank_wait times out.) */
/* Whole-lattice totals, ncclAllReduce over the ranks.  Blocking. */
int ising_rank_count(ising_ctx *ctx, uint64_t *up, uint64_t *down);
int ising_rank_bond_equal(ising_ctx *ctx, int64_t *A);

/* Where a ring slab's time goes around its exchanges (new; the reference synchronises every device after every colour and has
 * nothing to report, optimized/main.cu:1779-1805).  ising_exchange_stats_begin arms sampling of the next `max_exchanges` deep
 * exchanges of this slab (ballot ring slabs with ghost rows, exchange overlapped with the launches: the default schedule of
 * ising_ring_sweep / ising_rank_sweep); ising_exchange_stats_fetch waits for the slab's streams and reports, over the sampled
 * exchanges (timestamps of HIP events that ride on the launches' dispatch packets and on the comm stream):
 *   launch_ms        duration of the fused launch between two exchanges;
 *   exchange_ms      from "the launch's edge strips have finished their last level" (the exchange may start) to "the neighbours'
 *                    rows are in place" -- transport plus whatever the NEIGHBOURS were late by;
 *   go_after_end_ms  the exchange's end relative to the END of the launch whose rows it carries: negative = hidden in the launch's
 *                    tail, positive = the next launch had to wait that long (a slow link, or a neighbour that is behind);
 *   gap_ms           end of a launch to the start of the next (includes the positive part of go_after_end_ms).
 * Other schedules sample nothing (exchanges = 0).  Costs two events per exchange on the comm stream, nothing on the compute stream.
 * Round 6: over the peer (IPC) transport with a device per rank ONE persistent launch carries several exchange epochs (ISING_RING_EPOCHS; docs/SWITCHES.md), the
 * exchanges running next to it: `launches` = sampled launches (max_exchanges bounds these), `exchanges` = the exchange epochs inside them; launch_ms is a whole
 * launch, exchange_ms / go_after_end_ms / gap_ms describe each launch's LAST exchange -- the one the next launch waits for. */
typedef struct ising_exchange_stats {
	int32_t exchanges;
	float launch_ms_mean, launch_ms_max;
	float exchange_ms_mean, exchange_ms_max;
	float go_after_end_ms_mean, go_after_end_ms_max;
	float gap_ms_mean, gap_ms_max;
	int32_t launches;
} ising_exchange_stats;
int ising_exchange_stats_begin(ising_ctx *ctx, int max_exchanges);
int ising_exchange_stats_fetch(ising_ctx *ctx, ising_exchange_stats *out);

/* ---- a batch of independent lattices advancing together (the temperature-sweep driver; BASELINE config 5).  The
 * reference runs one lattice per process and temperature (optimized/main.cu:1596-1598, :1465-1471).  Contexts of one shape
 * (nslabs == 1, ballot layout -- ISING_LAYOUT_BALLOT or what AUTO picked --, no sub-lattices, no couplings) on one device and
 * one stream form a batch: ising_batch_sweep advances ALL of them by `nsweeps` sweeps (iterations first_it ..
 * first_it + nsweeps - 1, each lattice with its own seed and its own temperature as of ising_set_temperature) in fused
 * launches whose levels hold the tickets of every lattice, so that lattices too small to fill the chip with tall strips
 * (8192^2: 2766 flips/ns alone) run at the large-lattice rate; ising_batch_measure_enqueue adds ONE launch that takes the
 * up count and the bond sum (ising_count, ising_bond_equal) of every member.  Each member's spins are bit for bit what
 * ising_sweep on that context gives, and the members stay ordinary contexts (init, read, sweep them alone in between: same
 * stream).  Asynchronous like ising_sweep; the fetch blocks.
 * Round 6 -- lattices of the quad path (what ISING_LAYOUT_AUTO picks up to ~2^26 spins and eight blocks of 2048 columns; ising_sweep_info: 4) form batches
 * too, all members of one kind: ONE launch per pass carries the tiles of every member and the draws of every member for the pass to come, tile and pass length
 * chosen for the many tiles of the batch (ising_batch_quad_info), so that 31 x 2048^2 -- the finite-size end of a temperature series; the reference's own
 * many-small-systems mode, --xsl/--ysl, optimized/main.cu:1423-1457, knows one temperature -- run at the rate of one tall lattice.
 * ising_batch_sweep_counted is ising_sweep_counted for a batch: the print points ride inside the passes of a quad batch (a ballot batch: one measuring launch each). */
typedef struct ising_batch ising_batch;

int ising_batch_create(ising_ctx **ctxs, int n, ising_batch **out);
int ising_batch_destroy(ising_batch *b);                              /* before its members are destroyed */
int ising_batch_info(ising_batch *b, int *strip_rows, int *wg_per_cu, int *lattices); /* launch shape chosen for the batch */
int ising_batch_sweep(ising_batch *b, int first_it, int nsweeps);
int ising_batch_measure_enqueue(ising_batch *b);
/* up[k * lattices + r], bond_equal[...]: measurement k (enqueue order) of member r; at most 1024 measurements may be pending */
int ising_batch_measure_fetch(ising_batch *b, uint64_t *up, int64_t *bond_equal, int max_n, int *n);
/* `nsweeps` sweeps of every member (iterations first_it ..), the up spins -- and, bond_equal not NULL, ising_bond_equal's sums -- after every iteration that is a
 * multiple of `every`: up[k * lattices + r] = point k, member r (the reference's print points, optimized/main.cu:1785-1798, for every lattice of the batch).  Blocks. */
int ising_batch_sweep_counted(ising_batch *b, int first_it, int nsweeps, int every, uint64_t *up, int64_t *bond_equal, int max_counts, int *ncounts);
int ising_batch_quad_info(ising_batch *b, int *tile_row_groups, int *sweeps_per_pass, int *waves); /* a quad batch's shape (zeros: a ballot batch) */

/* -- one process per slab WITHOUT RCCL: direct peer access, the reference's own multi-GPU mechanism
 * (cudaDeviceCanAccessPeer / cudaDeviceEnablePeerAccess, optimized/main.cu:1496-1537; remote loads of the two rows outside
 * each slab, :1637-1642, loadTile :413-428), across processes.  Every rank exports a description of its slab's buffers
 * (hipIpcGetMemHandle of the spin and coupling arrays, the name of a 4 KiB POSIX shared-memory segment holding its flags),
 * the launcher's own channel (torch.distributed, MPI, a file) gathers the nslabs blobs in slab order, every rank attaches:
 * it maps the previous and the next slab's arrays (hipIpcOpenMemHandle; a neighbour in the same process is used directly)
 * and all ranks' flag segments.  From then on the ising_rank_* calls above run on ISING_TRANSPORT_IPC: a rank pushes its
 * first / last rows into its neighbours' halo or ghost rows with device-to-device copies on its comm stream and tells them
 * so through monotone epoch counters in their flag segments, which small kernels on the waiting streams poll -- no host
 * round trip inside a sweep, no stream or event shared between processes; totals (ising_rank_count) are summed by the
 * hosts through the same segments.  Ranks may share a device (several processes on one GPU: what a 1-GPU box can run of
 * an N-rank ring) or own one each (peer access over xGMI). */
#define ISING_IPC_BLOB_BYTES 256
int ising_ipc_export(ising_ctx *ctx, void *blob_out);                  /* ISING_IPC_BLOB_BYTES bytes describing this rank's slab */
int ising_ipc_attach(ising_ctx *ctx, const void *blobs, int nblobs);   /* nblobs == nslabs blobs in slab order; collective */

/* Two-point correlations, getCorr2D_k + computeCorr (optimized/main.cu:870-965, :1072-1138): for j = 1..ncorr
 * (ncorr <= 128 = MAX_CORR_LEN, :70)  sums[j-1] = sum over all sites of [s(r,c)==s(r,c+j) ? +1 : -1] +
 * [s(r,c)==s(r+j,c) ? +1 : -1], columns periodic in X, rows periodic in the whole lattice.  These are the exact
 * integers the reference accumulates in doubles; it then prints sums[j-1] / (2*X*Y*ndev) (:1131).  Each slab needs
 * at least ncorr rows.  With sub-lattices both wraps stay inside the site's own sub-lattice (getCorr2DRepl_k, :967-1070)
 * and YSL must be >= ncorr.  Blocking. */
int ising_correlations(ising_ctx *ctx, int ncorr, int64_t *sums);            /* nslabs == 1 */
int ising_ring_correlations(ising_ctx **ctxs, int n, int ncorr, int64_t *sums); /* totals over all slabs of a ring */

#ifdef __cplusplus
}
#endif
#endif /* ISING_HIP_H */
