// mhx_host_expand.h -- host side of the ACCEPT-COMPACTED return path (include/mhx.h: mhx_compact_hdr, mhx_compact_expand).
//
// What `sample` returns is a host container with one row per iteration (ext/AdvancedMHMCMCChainsExt.jl:12-39), and a rejected
// transition re-emits the previous Transition (src/mh-core.jl:109-114): at the acceptance rates Metropolis-Hastings is tuned to
// (0.234) three quarters of the tensor are byte-for-byte repeats of the row above.  The device therefore ships, per slab, a bit
// per (sample, chain) -- "this chain's column differs from the sample before" -- and the columns of the chains whose bit is set;
// the host threads here rebuild the caller's [n_samples][dim+1][nchains] tensor from that while the next slab is on the link.
// Pure host code (no HIP): the element type is only a width (4 or 8 bytes), so one object serves both instantiations of the engine.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include "../../include/mhx.h"

// One slab to expand.  `wait` (may be NULL) is called by ONE worker before the block is read -- the D2H copy that fills it is
// still in flight when the job is queued; a non-zero return fails the job (and every later one).
struct mhx_expand_job {
    const void* block = nullptr;         // mhx_compact_hdr + arrays + payload (page-locked staging memory)
    void* samples = nullptr;             // the caller's WHOLE tensor [n_samples][dim+1][nchains]
    uint8_t* accepted = nullptr;         // the caller's whole [n_samples][nchains], or NULL
    int64_t n_samples = 0;               // rows of the caller's tensor (bounds check of hdr.first_sample + hdr.count)
    int (*wait)(void*) = nullptr;
    void* wait_arg = nullptr;
};

class mhx_expander;
// threads <= 0: as many as the process may use (affinity mask, cgroup CPU quota), at most 64.  numa_node >= 0: the workers keep to
// the CPUs of that memory node (the GPU's: the blocks they read were written there by its DMA engine) where the process may run on any
mhx_expander* mhx_expander_create(int threads, int chunk_chains /* 0 = choose */, int numa_node /* -1 = anywhere */);

// ---- NUMA placement without libnuma (raw set_mempolicy; every call is a hint that may fail silently)
// memory node of a PCI device "dddd:bb:dd.f" (sysfs numa_node), -1 when unknown or the machine has one node
int mhx_numa_node_of_pci(const char* bus_id);
// the calling thread's NEW pages come from `node` if it has room (MPOL_PREFERRED) / from wherever the thread runs (default) again
void mhx_numa_prefer(int node);
void mhx_numa_default(void);
void mhx_expander_destroy(mhx_expander* e);
int mhx_expander_threads(const mhx_expander* e);
// queue a slab; jobs are expanded strictly in order (slab c + 1 reads the last row slab c wrote).  Returns its sequence number.
uint64_t mhx_expander_submit(mhx_expander* e, const mhx_expand_job& job);
// block until job `seq` (and all before it) has been expanded or has failed
void mhx_expander_wait(mhx_expander* e, uint64_t seq);
// block until nothing is queued or running; MHX_OK or the first error since the last drain (message via mhx_last_error)
int mhx_expander_drain(mhx_expander* e);
// seconds the workers spent between taking a block and finishing it, summed over the jobs since the last reset
double mhx_expander_busy_seconds(mhx_expander* e, int reset);

// how many threads this process may keep busy: min(hardware threads, affinity mask, cgroup v2 / v1 CPU quota), >= 1
int mhx_host_usable_cpus(void);
// bytes of the fixed part of a block (header + mask + rank + accepted arrays, each padded to 8 bytes) = hdr.payload_offset
size_t mhx_compact_payload_offset(uint32_t count, uint32_t words, uint32_t nchains);
