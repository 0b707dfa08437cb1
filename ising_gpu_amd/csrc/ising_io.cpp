// ising_io.cpp -- the slab's boundary formats: the reference's packed rows (ising_read_packed / ising_write_packed, the
// D2H copy of dumpLattice, optimized/main.cu:1150-1152), the text dump (dumpLattice :1140-1209) and a binary checkpoint
// the reference does not have ("so 10^5-sweep runs can resume").
//
// Whatever the device layout, rows are converted ON THE DEVICE in bounded chunks (two staging buffers of at most 32 + 8 MiB):
//   ballot -> dense-order rows (ballot_to_dense_k) -> packed nibbles (dense_to_packed_k) -> host
// so a 65536^2 slab is neither expanded by a scalar host loop nor held twice -- in host memory or on the device.
#include "ising_ctx.hpp"

#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

using ising_host::bind;
using ising_host::fail;

namespace {

constexpr size_t STAGE_BYTES = 32u << 20;

int check_rows(ising_ctx *c, int color, int64_t row0, int64_t nrows, const void *host) {
	if (!c || !host) return fail(ISING_E_ARG, "null argument");
	if (color != ISING_BLACK && color != ISING_WHITE) return fail(ISING_E_ARG, "bad colour %d", color);
	if (row0 < 0 || nrows < 0 || row0 + nrows > c->cfg.Y) return fail(ISING_E_ARG, "rows [%lld,%lld) outside slab of %d rows", (long long)row0, (long long)(row0 + nrows), c->cfg.Y);
	return ISING_OK;
}

// rows per chunk such that the packed form of a chunk fits the staging buffer; allocates the buffer
int stage(ising_ctx *c, int64_t *chunk_rows) {
	const size_t row_bytes = (size_t)c->lld_packed * sizeof(uint64_t);
	int64_t rows = (int64_t)(STAGE_BYTES / row_bytes);
	if (rows < 1) rows = 1;
	const size_t words = (size_t)rows * c->lld_packed;
	if (c->pack_words < words) {
		if (c->d_pack) HIP_TRY(hipFree(c->d_pack));
		c->d_pack = nullptr;
		c->pack_words = 0;
		HIP_TRY(hipMalloc((void **)&c->d_pack, words * sizeof(uint64_t)));
		c->pack_words = words;
	}
	// ballot layout: the chunk in dense order on its way between the slab's words and the packed form (a quarter of the size)
	const size_t cwords = (size_t)rows * c->lld_dense;
	if (c->ballot && c->conv_words < cwords) {
		if (c->d_conv) HIP_TRY(hipFree(c->d_conv));
		c->d_conv = nullptr;
		c->conv_words = 0;
		HIP_TRY(hipMalloc((void **)&c->d_conv, cwords * sizeof(uint64_t)));
		c->conv_words = cwords;
	}
	*chunk_rows = rows;
	return ISING_OK;
}

// ballot layout: rows [r, r + nr) of `color` between the slab and the dense-order staging chunk
int ballot_chunk(ising_ctx *c, int color, int64_t r, int64_t nr, bool to_dense) {
	uint64_t *lat = c->lat(color) + r * c->lld;
	if (to_dense) HIP_TRY(ising::launch_ballot_to_dense(lat, reinterpret_cast<uint32_t *>(c->d_conv), c->gx, nr, c->stream));
	else HIP_TRY(ising::launch_dense_to_ballot(reinterpret_cast<const uint32_t *>(c->d_conv), lat, c->gx, nr, c->stream));
	return ISING_OK;
}

enum Format { PACKED, BITS }; // 64-bit words of 16 nibbles (the reference's) / 32-bit words of 32 spins (one per reference vector)

// Rows [row0, row0 + nrows) of one colour to host memory.  PACKED: nrows * X/32 uint64; BITS: nrows * X/64 uint32.
int read_rows(ising_ctx *c, int color, int64_t row0, int64_t nrows, void *host, Format fmt) {
	if (int rc = bind(c)) return rc;
	const size_t nvec_row = (size_t)c->lld_packed / 2; // reference vectors (= dense 32-bit words) per row
	if (!c->dense && fmt == PACKED) { // the device holds the boundary format itself
		HIP_TRY(hipMemcpyAsync(host, c->lat(color) + (size_t)row0 * c->lld, (size_t)nrows * c->lld * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
		HIP_TRY(hipStreamSynchronize(c->stream));
		return ISING_OK;
	}
	int64_t chunk = 0;
	if (int rc = stage(c, &chunk)) return rc;
	for (int64_t r = row0; r < row0 + nrows; r += chunk) {
		const int64_t nr = std::min(chunk, row0 + nrows - r);
		const size_t nvec = (size_t)nr * nvec_row;
		const uint32_t *bits = nullptr; // dense-order words of the chunk on the device
		if (c->ballot) {
			if (int rc = ballot_chunk(c, color, r, nr, true)) return rc;
			bits = reinterpret_cast<const uint32_t *>(c->d_conv);
		} else if (c->dense) {
			bits = reinterpret_cast<const uint32_t *>(c->lat(color) + (size_t)r * c->lld);
		} else { // nibble layout, BITS wanted
			HIP_TRY(ising::launch_packed_to_dense(c->lat(color) + (size_t)r * c->lld, reinterpret_cast<uint32_t *>(c->d_pack), nvec, c->stream));
			bits = reinterpret_cast<const uint32_t *>(c->d_pack);
		}
		if (fmt == PACKED) {
			HIP_TRY(ising::launch_dense_to_packed(bits, c->d_pack, nvec, c->stream));
			HIP_TRY(hipMemcpyAsync(static_cast<uint64_t *>(host) + (size_t)(r - row0) * c->lld_packed, c->d_pack, nvec * 16, hipMemcpyDeviceToHost, c->stream));
		} else {
			HIP_TRY(hipMemcpyAsync(static_cast<uint32_t *>(host) + (size_t)(r - row0) * nvec_row, bits, nvec * 4, hipMemcpyDeviceToHost, c->stream));
		}
		HIP_TRY(hipStreamSynchronize(c->stream)); // the staging buffer is reused by the next chunk
	}
	return ISING_OK;
}

int write_rows(ising_ctx *c, int color, int64_t row0, int64_t nrows, const void *host, Format fmt) {
	if (color == ISING_BLACK || color == ISING_WHITE) c->ghost_depth[color] = 0; // (the ring's ghost rows of this colour are stale now)
	if (int rc = bind(c)) return rc;
	const size_t nvec_row = (size_t)c->lld_packed / 2;
	const size_t row_bytes = (size_t)c->lld * sizeof(uint64_t); // device row
	if (!c->dense && fmt == PACKED) {
		HIP_TRY(hipMemcpyAsync(c->lat(color) + (size_t)row0 * c->lld, host, (size_t)nrows * row_bytes, hipMemcpyHostToDevice, c->stream));
	} else {
		int64_t chunk = 0;
		if (int rc = stage(c, &chunk)) return rc;
		for (int64_t r = row0; r < row0 + nrows; r += chunk) {
			const int64_t nr = std::min(chunk, row0 + nrows - r);
			const size_t nvec = (size_t)nr * nvec_row;
			// where the chunk's dense-order words go: the slab itself (dense), the dense-order image (ballot), staging (nibble)
			uint32_t *bits = c->ballot ? reinterpret_cast<uint32_t *>(c->d_conv)
			                 : (c->dense ? reinterpret_cast<uint32_t *>(c->lat(color) + (size_t)r * c->lld) : nullptr);
			if (fmt == PACKED) { // 1 bit/spin layouts
				HIP_TRY(hipMemcpyAsync(c->d_pack, static_cast<const uint64_t *>(host) + (size_t)(r - row0) * c->lld_packed, nvec * 16, hipMemcpyHostToDevice, c->stream));
				HIP_TRY(ising::launch_packed_to_dense(c->d_pack, bits, nvec, c->stream));
			} else if (bits) {
				HIP_TRY(hipMemcpyAsync(bits, static_cast<const uint32_t *>(host) + (size_t)(r - row0) * nvec_row, nvec * 4, hipMemcpyHostToDevice, c->stream));
			} else { // BITS into the nibble layout
				uint32_t *st = reinterpret_cast<uint32_t *>(c->d_pack);
				HIP_TRY(hipMemcpyAsync(st, static_cast<const uint32_t *>(host) + (size_t)(r - row0) * nvec_row, nvec * 4, hipMemcpyHostToDevice, c->stream));
				HIP_TRY(ising::launch_dense_to_packed(st, c->lat(color) + (size_t)r * c->lld, nvec, c->stream));
			}
			if (c->ballot) if (int rc = ballot_chunk(c, color, r, nr, false)) return rc;
			HIP_TRY(hipStreamSynchronize(c->stream));
		}
	}
	// a single slab that wraps in place: the halo rows mirror the opposite edge rows (same device layout)
	if (c->wrap && nrows > 0) {
		uint64_t *base = c->lat(color);
		if (row0 == 0) HIP_TRY(hipMemcpyAsync(base + c->color_words, base, row_bytes, hipMemcpyDeviceToDevice, c->stream));
		if (row0 + nrows == c->cfg.Y) HIP_TRY(hipMemcpyAsync(base - c->lld, base + c->color_words - c->lld, row_bytes, hipMemcpyDeviceToDevice, c->stream));
	}
	HIP_TRY(hipStreamSynchronize(c->stream));
	return ISING_OK;
}

// ---- checkpoint file: header, then ALL black rows of the lattice in global row order, then all white rows (BITS
// format, X/64 uint32 per row: 1 bit per spin), then the number of up spins as a check.  Global row order makes the
// file independent of the slab decomposition it was written from.
struct CheckpointHeader {
	char magic[8];          // "ISNGCKP1"
	uint32_t header_bytes;  // sizeof(CheckpointHeader)
	uint32_t encoding;      // 1: BITS
	int32_t X, Y_total;
	int32_t nslabs_written; // informational
	int32_t XSL, YSL, use_J;
	uint32_t temp_bits, J_prob_bits;
	uint64_t seed;
	int64_t it;             // completed sweeps; the next sweep is iteration it + 1 (optimized/main.cu:1766: j + 1)
	uint64_t payload_bytes;
	uint64_t reserved[8];
};
static_assert(sizeof(CheckpointHeader) == 136, "checkpoint header layout");
const char CKPT_MAGIC[8] = {'I', 'S', 'N', 'G', 'C', 'K', 'P', '1'};

int read_header(FILE *fp, const char *path, CheckpointHeader *h) {
	if (fread(h, sizeof(*h), 1, fp) != 1) return fail(ISING_E_IO, "%s: short read of the checkpoint header", path);
	if (memcmp(h->magic, CKPT_MAGIC, 8) || h->header_bytes != sizeof(*h) || h->encoding != 1)
		return fail(ISING_E_IO, "%s is not an ising checkpoint (or a newer format)", path);
	if (h->X <= 0 || h->Y_total <= 0 || (h->X % 2048) || (h->Y_total % 16) || h->payload_bytes != 2ull * h->Y_total * (uint64_t)(h->X / 64) * 4)
		return fail(ISING_E_IO, "%s: inconsistent checkpoint header", path);
	return ISING_OK;
}

} // namespace

extern "C" {

int ising_read_packed(ising_ctx *c, int color, int64_t row0, int64_t nrows, uint64_t *dst_host) {
	if (int rc = check_rows(c, color, row0, nrows, dst_host)) return rc;
	return read_rows(c, color, row0, nrows, dst_host, PACKED);
}

int ising_write_packed(ising_ctx *c, int color, int64_t row0, int64_t nrows, const uint64_t *src_host) {
	if (int rc = check_rows(c, color, row0, nrows, src_host)) return rc;
	return write_rows(c, color, row0, nrows, src_host, PACKED);
}

int ising_read_bits(ising_ctx *c, int color, int64_t row0, int64_t nrows, uint32_t *dst_host) {
	if (int rc = check_rows(c, color, row0, nrows, dst_host)) return rc;
	return read_rows(c, color, row0, nrows, dst_host, BITS);
}

int ising_write_bits(ising_ctx *c, int color, int64_t row0, int64_t nrows, const uint32_t *src_host) {
	if (int rc = check_rows(c, color, row0, nrows, src_host)) return rc;
	return write_rows(c, color, row0, nrows, src_host, BITS);
}

int ising_dump_text(ising_ctx *c, const char *prefix) {
	if (!c || !prefix) return fail(ISING_E_ARG, "null argument");
	if (int rc = bind(c)) return rc;
	char fname[512];
	snprintf(fname, sizeof(fname), "%s%d.txt", prefix, c->cfg.slab); // optimized/main.cu:1157,:1185
	FILE *fp = fopen(fname, "w");
	if (!fp) return fail(ISING_E_IO, "cannot open %s for writing: %s", fname, strerror(errno));
	const int lp = c->lld_packed;
	const int64_t chunk = std::max<int64_t>(1, (int64_t)(STAGE_BYTES / ((size_t)lp * 8)));
	std::vector<uint64_t> hb((size_t)chunk * lp), hw((size_t)chunk * lp);
	std::string line((size_t)c->cfg.X + 1, '\n');
	int rc = ISING_OK;
	for (int64_t r0 = 0; r0 < c->cfg.Y && rc == ISING_OK; r0 += chunk) {
		const int64_t nr = std::min<int64_t>(chunk, c->cfg.Y - r0);
		rc = read_rows(c, ISING_BLACK, r0, nr, hb.data(), PACKED);
		if (rc == ISING_OK) rc = read_rows(c, ISING_WHITE, r0, nr, hw.data(), PACKED);
		for (int64_t i = 0; i < nr && rc == ISING_OK; i++) {
			char *q = &line[0];
			const uint64_t *b = hb.data() + (size_t)i * lp, *w = hw.data() + (size_t)i * lp;
			// local row parity decides the interleave, as in the reference's per-device loop (optimized/main.cu:1188-1201);
			// only bit 0 of a nibble is ever set, so the hex digit of a spin is '0' or '1'
			const bool odd = (r0 + i) & 1;
			for (int j = 0; j < lp; j++) {
				const uint64_t first = odd ? w[j] : b[j], second = odd ? b[j] : w[j];
				for (int k = 0; k < 64; k += 4) {
					*q++ = (char)('0' + ((first >> k) & 0xF));
					*q++ = (char)('0' + ((second >> k) & 0xF));
				}
			}
			if (fwrite(line.data(), 1, line.size(), fp) != line.size()) rc = fail(ISING_E_IO, "write to %s failed: %s", fname, strerror(errno));
		}
	}
	if (fclose(fp) != 0 && rc == ISING_OK) rc = fail(ISING_E_IO, "closing %s failed: %s", fname, strerror(errno));
	return rc;
}

// ------------------------------------------------------------------------------------------------ checkpoint
int ising_checkpoint_info_read(const char *path, ising_checkpoint_info *info) {
	if (!path || !info) return fail(ISING_E_ARG, "null argument");
	FILE *fp = fopen(path, "rb");
	if (!fp) return fail(ISING_E_IO, "cannot open %s: %s", path, strerror(errno));
	CheckpointHeader h;
	const int rc = read_header(fp, path, &h);
	fclose(fp);
	if (rc) return rc;
	info->X = h.X;
	info->Y_total = h.Y_total;
	info->nslabs_written = h.nslabs_written;
	info->XSL = h.XSL;
	info->YSL = h.YSL;
	info->use_J = h.use_J;
	memcpy(&info->temp, &h.temp_bits, 4);
	memcpy(&info->J_prob, &h.J_prob_bits, 4);
	info->seed = h.seed;
	info->it = h.it;
	return ISING_OK;
}

int ising_ring_checkpoint_save(ising_ctx **ctxs, int n, const char *path, int64_t it) {
	if (!ctxs || n < 1 || !path) return fail(ISING_E_ARG, "bad argument");
	for (int k = 0; k < n; k++)
		if (!ctxs[k] || ctxs[k]->cfg.slab != k || ctxs[k]->cfg.nslabs != n) return fail(ISING_E_ARG, "ring slot %d does not hold slab %d of %d", k, k, n);
	if (int rc = ising_ring_synchronize(ctxs, n)) return rc;
	const ising_ctx *c0 = ctxs[0];
	CheckpointHeader h;
	memset(&h, 0, sizeof(h));
	memcpy(h.magic, CKPT_MAGIC, 8);
	h.header_bytes = sizeof(h);
	h.encoding = 1;
	h.X = c0->cfg.X;
	h.Y_total = c0->cfg.Y * n;
	h.nslabs_written = n;
	h.XSL = c0->cfg.XSL;
	h.YSL = c0->cfg.YSL;
	h.use_J = c0->cfg.use_J;
	memcpy(&h.temp_bits, &c0->cfg.temp, 4);
	memcpy(&h.J_prob_bits, &c0->cfg.J_prob, 4);
	h.seed = c0->cfg.seed;
	h.it = it;
	const size_t row_words = (size_t)c0->cfg.X / 64; // uint32 per row
	h.payload_bytes = 2ull * h.Y_total * row_words * 4;
	const std::string tmp = std::string(path) + ".part";
	FILE *fp = fopen(tmp.c_str(), "wb");
	if (!fp) return fail(ISING_E_IO, "cannot open %s for writing: %s", tmp.c_str(), strerror(errno));
	int rc = fwrite(&h, sizeof(h), 1, fp) == 1 ? ISING_OK : fail(ISING_E_IO, "write to %s failed", tmp.c_str());
	const int64_t chunk = std::max<int64_t>(1, (int64_t)(STAGE_BYTES / (row_words * 4)));
	std::vector<uint32_t> buf((size_t)std::min<int64_t>(chunk, c0->cfg.Y) * row_words);
	uint64_t up_total = 0;
	for (int color = 0; color < 2 && rc == ISING_OK; color++) {
		for (int k = 0; k < n && rc == ISING_OK; k++) {
			for (int64_t r = 0; r < c0->cfg.Y && rc == ISING_OK; r += chunk) {
				const int64_t nr = std::min<int64_t>(chunk, c0->cfg.Y - r);
				rc = read_rows(ctxs[k], color, r, nr, buf.data(), BITS);
				if (rc != ISING_OK) break;
				const size_t nw = (size_t)nr * row_words;
				for (size_t i = 0; i < nw; i++) up_total += (uint64_t)__builtin_popcount(buf[i]);
				if (fwrite(buf.data(), 4, nw, fp) != nw) rc = fail(ISING_E_IO, "write to %s failed: %s", tmp.c_str(), strerror(errno));
			}
		}
	}
	if (rc == ISING_OK && fwrite(&up_total, sizeof(up_total), 1, fp) != 1) rc = fail(ISING_E_IO, "write to %s failed", tmp.c_str());
	if (fclose(fp) != 0 && rc == ISING_OK) rc = fail(ISING_E_IO, "closing %s failed: %s", tmp.c_str(), strerror(errno));
	if (rc == ISING_OK && rename(tmp.c_str(), path) != 0) rc = fail(ISING_E_IO, "cannot rename %s to %s: %s", tmp.c_str(), path, strerror(errno));
	if (rc != ISING_OK) remove(tmp.c_str());
	return rc;
}

int ising_ring_checkpoint_load(ising_ctx **ctxs, int n, const char *path, int64_t *it) {
	if (!ctxs || n < 1 || !path) return fail(ISING_E_ARG, "bad argument");
	for (int k = 0; k < n; k++)
		if (!ctxs[k] || ctxs[k]->cfg.slab != k || ctxs[k]->cfg.nslabs != n) return fail(ISING_E_ARG, "ring slot %d does not hold slab %d of %d", k, k, n);
	if (int rc = ising_ring_synchronize(ctxs, n)) return rc;
	const ising_ctx *c0 = ctxs[0];
	FILE *fp = fopen(path, "rb");
	if (!fp) return fail(ISING_E_IO, "cannot open %s: %s", path, strerror(errno));
	CheckpointHeader h;
	int rc = read_header(fp, path, &h);
	if (rc == ISING_OK && (h.X != c0->cfg.X || h.Y_total != c0->cfg.Y * n))
		rc = fail(ISING_E_ARG, "%s holds a %d x %d lattice, the ring is %d x %d", path, h.Y_total, h.X, c0->cfg.Y * n, c0->cfg.X);
	// the Philox streams continue only under the seed (and sub-lattice / coupling setup) the file was written with
	if (rc == ISING_OK && h.seed != c0->cfg.seed) rc = fail(ISING_E_ARG, "%s was written with seed %llu, the ring uses %llu", path, (unsigned long long)h.seed, (unsigned long long)c0->cfg.seed);
	if (rc == ISING_OK && (h.XSL != c0->cfg.XSL || h.YSL != c0->cfg.YSL || h.use_J != c0->cfg.use_J || (h.use_J && memcmp(&h.J_prob_bits, &c0->cfg.J_prob, 4))))
		rc = fail(ISING_E_ARG, "%s was written with other sub-lattice / coupling settings", path);
	const size_t row_words = (size_t)c0->cfg.X / 64;
	const int64_t chunk = std::max<int64_t>(1, (int64_t)(STAGE_BYTES / (row_words * 4)));
	std::vector<uint32_t> buf((size_t)std::min<int64_t>(chunk, c0->cfg.Y) * row_words);
	uint64_t up_total = 0, up_file = 0;
	for (int color = 0; color < 2 && rc == ISING_OK; color++) {
		for (int k = 0; k < n && rc == ISING_OK; k++) {
			for (int64_t r = 0; r < c0->cfg.Y && rc == ISING_OK; r += chunk) {
				const int64_t nr = std::min<int64_t>(chunk, c0->cfg.Y - r);
				const size_t nw = (size_t)nr * row_words;
				if (fread(buf.data(), 4, nw, fp) != nw) { rc = fail(ISING_E_IO, "%s: short read", path); break; }
				for (size_t i = 0; i < nw; i++) up_total += (uint64_t)__builtin_popcount(buf[i]);
				rc = write_rows(ctxs[k], color, r, nr, buf.data(), BITS);
			}
		}
	}
	if (rc == ISING_OK && fread(&up_file, sizeof(up_file), 1, fp) != 1) rc = fail(ISING_E_IO, "%s: short read", path);
	fclose(fp);
	if (rc == ISING_OK && up_file != up_total) rc = fail(ISING_E_IO, "%s is damaged: %llu up spins read, %llu recorded", path, (unsigned long long)up_total, (unsigned long long)up_file);
	if (rc == ISING_OK) {
		uint64_t up = 0, down = 0;
		rc = ising_ring_count(ctxs, n, &up, &down);
		if (rc == ISING_OK && up != up_file) rc = fail(ISING_E_STATE, "checkpoint load: the device holds %llu up spins, the file %llu", (unsigned long long)up, (unsigned long long)up_file);
	}
	if (rc == ISING_OK && it) *it = h.it;
	return rc;
}

// ---- the same file from a ring of PROCESSES (ising_rank_*: one slab per process).  Global row order makes every rank's rows a
// contiguous range per colour: rank 0 lays the file out (header, size, the total's trailer), every rank writes / reads its own
// two ranges in place, the collective count is the barrier between the phases and the check at the end.  The path must name
// the same file for every rank (one node, or a shared file system).
static int rank_header(const ising_ctx *c, int64_t it, CheckpointHeader *h) {
	memset(h, 0, sizeof(*h));
	memcpy(h->magic, CKPT_MAGIC, 8);
	h->header_bytes = sizeof(*h);
	h->encoding = 1;
	h->X = c->cfg.X;
	h->Y_total = c->cfg.Y * c->cfg.nslabs;
	h->nslabs_written = c->cfg.nslabs;
	h->XSL = c->cfg.XSL;
	h->YSL = c->cfg.YSL;
	h->use_J = c->cfg.use_J;
	memcpy(&h->temp_bits, &c->cfg.temp, 4);
	memcpy(&h->J_prob_bits, &c->cfg.J_prob, 4);
	h->seed = c->cfg.seed;
	h->it = it;
	h->payload_bytes = 2ull * h->Y_total * (uint64_t)(c->cfg.X / 64) * 4;
	return ISING_OK;
}

int ising_rank_checkpoint_save(ising_ctx *c, const char *path, int64_t it) {
	if (!c || !path) return fail(ISING_E_ARG, "null argument");
	if (!c->rank_mode) return fail(ISING_E_STATE, "the slab is not attached to a multi-process ring (single-process rings: ising_ring_checkpoint_save)");
	// (a rank on which one of these fails still enters every collective below -- its failure is folded into `rc` and agreed
	// on with the others -- instead of returning early and leaving them waiting in an all-reduce)
	int rc = ising_rank_wait(c, -1);
	uint64_t up = 0, down = 0;
	if (const int rc2 = ising_rank_count(c, &up, &down); rc == ISING_OK) rc = rc2; // the total for the trailer; every rank has stopped sweeping
	CheckpointHeader h;
	rank_header(c, it, &h);
	const std::string tmp = std::string(path) + ".part";
	const size_t row_bytes = (size_t)c->cfg.X / 64 * 4;
	if (c->cfg.slab == 0 && rc == ISING_OK) { // lay the file out
		FILE *fp = fopen(tmp.c_str(), "wb");
		if (!fp) rc = fail(ISING_E_IO, "cannot open %s for writing: %s", tmp.c_str(), strerror(errno));
		if (rc == ISING_OK && fwrite(&h, sizeof(h), 1, fp) != 1) rc = fail(ISING_E_IO, "write to %s failed", tmp.c_str());
		if (rc == ISING_OK && (fseeko(fp, (off_t)(sizeof(h) + h.payload_bytes), SEEK_SET) != 0 || fwrite(&up, sizeof(up), 1, fp) != 1)) rc = fail(ISING_E_IO, "cannot size %s: %s", tmp.c_str(), strerror(errno));
		if (fp && fclose(fp) != 0 && rc == ISING_OK) rc = fail(ISING_E_IO, "closing %s failed: %s", tmp.c_str(), strerror(errno));
	}
	// the ranks agree on every stage's outcome (a sum of failure flags, which is also their barrier): nobody writes into a file
	// that could not be laid out, and a file some rank could not fill never gets its name
	unsigned long long bad = 0;
	if (int rc2 = ising_host::rank_sum_u64(c, rc != ISING_OK, &bad)) return rc2; // (the file exists)
	if (bad && rc == ISING_OK) rc = fail(ISING_E_IO, "rank 0 could not create %s", tmp.c_str());
	if (rc == ISING_OK) {
		FILE *fp = fopen(tmp.c_str(), "r+b");
		if (!fp) rc = fail(ISING_E_IO, "cannot open %s: %s", tmp.c_str(), strerror(errno));
		const int64_t chunk = std::max<int64_t>(1, (int64_t)(STAGE_BYTES / row_bytes));
		std::vector<uint32_t> buf((size_t)std::min<int64_t>(chunk, c->cfg.Y) * (row_bytes / 4));
		for (int color = 0; color < 2 && rc == ISING_OK; color++) {
			const off_t base = (off_t)(sizeof(h) + ((uint64_t)color * h.Y_total + (uint64_t)c->cfg.slab * c->cfg.Y) * row_bytes);
			for (int64_t r = 0; r < c->cfg.Y && rc == ISING_OK; r += chunk) {
				const int64_t nr = std::min<int64_t>(chunk, c->cfg.Y - r);
				rc = read_rows(c, color, r, nr, buf.data(), BITS);
				if (rc == ISING_OK && (fseeko(fp, base + (off_t)((uint64_t)r * row_bytes), SEEK_SET) != 0 || fwrite(buf.data(), row_bytes, (size_t)nr, fp) != (size_t)nr))
					rc = fail(ISING_E_IO, "write to %s failed: %s", tmp.c_str(), strerror(errno));
			}
		}
		if (fp && fclose(fp) != 0 && rc == ISING_OK) rc = fail(ISING_E_IO, "closing %s failed: %s", tmp.c_str(), strerror(errno));
	}
	if (int rc2 = ising_host::rank_sum_u64(c, rc != ISING_OK, &bad)) return rc2; // (every rank's rows are in the file)
	if (bad && rc == ISING_OK) rc = fail(ISING_E_IO, "%llu rank(s) could not write their rows into %s", bad, tmp.c_str());
	if (c->cfg.slab == 0) {
		if (rc == ISING_OK && rename(tmp.c_str(), path) != 0) rc = fail(ISING_E_IO, "cannot rename %s to %s: %s", tmp.c_str(), path, strerror(errno));
		else if (rc != ISING_OK) (void)remove(tmp.c_str());
	}
	if (int rc2 = ising_host::rank_sum_u64(c, rc != ISING_OK, &bad)) return rc2; // (the file has its name)
	if (bad && rc == ISING_OK) rc = fail(ISING_E_IO, "rank 0 could not give %s its name", path);
	return rc;
}

int ising_rank_checkpoint_load(ising_ctx *c, const char *path, int64_t *it) {
	if (!c || !path) return fail(ISING_E_ARG, "null argument");
	if (!c->rank_mode) return fail(ISING_E_STATE, "the slab is not attached to a multi-process ring (single-process rings: ising_ring_checkpoint_load)");
	// (a rank that fails still takes part in the collectives below: the ranks stay in step and all of them report the failure)
	int rc = ising_rank_wait(c, -1);
	FILE *fp = fopen(path, "rb");
	if (!fp && rc == ISING_OK) rc = fail(ISING_E_IO, "cannot open %s: %s", path, strerror(errno));
	CheckpointHeader h;
	memset(&h, 0, sizeof(h));
	if (rc == ISING_OK) rc = read_header(fp, path, &h);
	if (rc == ISING_OK && (h.X != c->cfg.X || h.Y_total != c->cfg.Y * c->cfg.nslabs))
		rc = fail(ISING_E_ARG, "%s holds a %d x %d lattice, the ring is %d x %d", path, h.Y_total, h.X, c->cfg.Y * c->cfg.nslabs, c->cfg.X);
	if (rc == ISING_OK && h.seed != c->cfg.seed) rc = fail(ISING_E_ARG, "%s was written with seed %llu, the ring uses %llu", path, (unsigned long long)h.seed, (unsigned long long)c->cfg.seed);
	if (rc == ISING_OK && (h.XSL != c->cfg.XSL || h.YSL != c->cfg.YSL || h.use_J != c->cfg.use_J || (h.use_J && memcmp(&h.J_prob_bits, &c->cfg.J_prob, 4))))
		rc = fail(ISING_E_ARG, "%s was written with other sub-lattice / coupling settings", path);
	const size_t row_bytes = (size_t)c->cfg.X / 64 * 4;
	const int64_t chunk = std::max<int64_t>(1, (int64_t)(STAGE_BYTES / row_bytes));
	std::vector<uint32_t> buf((size_t)std::min<int64_t>(chunk, c->cfg.Y) * (row_bytes / 4));
	for (int color = 0; color < 2 && rc == ISING_OK; color++) {
		const off_t base = (off_t)(sizeof(h) + ((uint64_t)color * h.Y_total + (uint64_t)c->cfg.slab * c->cfg.Y) * row_bytes);
		for (int64_t r = 0; r < c->cfg.Y && rc == ISING_OK; r += chunk) {
			const int64_t nr = std::min<int64_t>(chunk, c->cfg.Y - r);
			if (fseeko(fp, base + (off_t)((uint64_t)r * row_bytes), SEEK_SET) != 0 || fread(buf.data(), row_bytes, (size_t)nr, fp) != (size_t)nr) { rc = fail(ISING_E_IO, "%s: short read", path); break; }
			rc = write_rows(c, color, r, nr, buf.data(), BITS);
		}
	}
	uint64_t up_file = 0;
	if (rc == ISING_OK && (fseeko(fp, (off_t)(sizeof(h) + h.payload_bytes), SEEK_SET) != 0 || fread(&up_file, sizeof(up_file), 1, fp) != 1)) rc = fail(ISING_E_IO, "%s: short read", path);
	if (fp) fclose(fp);
	uint64_t up = 0, down = 0;
	if (int rc2 = ising_rank_count(c, &up, &down)) return rc2;
	if (rc == ISING_OK && up != up_file) rc = fail(ISING_E_IO, "%s is damaged or was not read whole: the ring holds %llu up spins, the file records %llu", path, (unsigned long long)up, (unsigned long long)up_file);
	unsigned long long bad = 0;
	if (int rc2 = ising_host::rank_sum_u64(c, rc != ISING_OK, &bad)) return rc2;
	if (bad && rc == ISING_OK) rc = fail(ISING_E_IO, "%llu rank(s) could not load their rows from %s: the ring's state is undefined", bad, path);
	if (rc == ISING_OK && it) *it = h.it;
	return rc;
}

} // extern "C"
