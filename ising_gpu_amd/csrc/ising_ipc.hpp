// ising_ipc.hpp -- what the slab ring (ising_ring.cpp) uses of the multi-process peer transport (ising_ipc.cpp), and the
// one-lane counter kernels both share.  Internal to the library.
#pragma once
#include "ising_ctx.hpp"

namespace ising_ipc {

bool attached(const ising_ctx *c);
// ranks of the attached ring (this one included) whose slabs live on this rank's device (1: the rank has its device to itself)
int sharing(const ising_ctx *c);
// pinned, device-visible word this rank's polling kernels look at: set by the host, they give up (NULL: not exported)
const uint32_t *abort_word(const ising_ctx *c);
void set_abort(ising_ctx *c, bool on);
// stream `s` waits until both neighbours have delivered the current epoch of `plane` into this slab's halo / ghost rows
int wait_plane(ising_ctx *c, int plane, hipStream_t s);
// deep exchange: everything stream `s` (default: the compute stream) holds so far is done with this slab's ghost rows of
// `color` once the stream gets here -- the neighbours may overwrite them with the next epoch
int release_ghosts(ising_ctx *c, int color, hipStream_t s = nullptr);
// push `bytes` from `first` / `last` (this slab's first / last rows of plane `color`) into the previous / next rank's rows at
// the place `halo_bot` / `halo_top` have in THIS slab's arrays (every slab of a ring has one shape), on the comm stream, and
// move the neighbours' READY counters behind the copies; depth > 1: wait for their FREE counters first
int push_rows(ising_ctx *c, int color, int depth, const void *first, const void *last, const void *halo_top, const void *halo_bot, size_t bytes);
// sum of one 64-bit value over all ranks, by the hosts, through the shared segments
int allreduce_u64(ising_ctx *c, unsigned long long mine_val, unsigned long long *out, int timeout_ms = 120000);
// unmaps the neighbours and removes this rank's segment (its counters belong to one attachment)
void destroy(ising_ctx *c);

// one lane waits until a counter in device memory has reached `need` / writes it, behind everything the stream did before
int counter_wait_on(hipStream_t s, const uint32_t *counter, uint32_t need, const uint32_t *abort_flag);
int counter_set_on(hipStream_t s, uint32_t *counter, uint32_t value);

} // namespace ising_ipc
