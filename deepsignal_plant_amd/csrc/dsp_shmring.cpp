// dsp_shmring.cpp -- a ring of text blocks in POSIX shared memory: ONE producer per node, the node's ranks as consumers.
//
// Why: a feature file written by the reference's `extract --gzip` is one gzip stream (extract_features.py writers, read
// back with gzip.open at call_modifications.py:66-69).  A deflate stream cannot be entered in the middle, so with N ranks
// per node every rank used to inflate the whole file (N x the single-threaded zlib work, N x the file reads).  Now the
// node's first rank inflates once, straight into the slots of this ring, as blocks of complete rows; block i belongs to
// rank i % world, which copies it out and frees the slot.  (Files this build writes are BGZF and never come here: their
// members are dealt to ranks by range and inflated in parallel, dsp_gz.cpp.)
//
// Layout of the segment: Header | n_slots x SlotHeader | n_slots x slot_bytes of payload.  All synchronisation is by
// C++11 atomics in the mapping (lock-free 64-bit atomics are address-free on x86-64) and short sleeps: a block is tens of
// MB and takes > 100 ms to inflate, so a 50 us poll costs nothing.
#include "dsp_amd.h"

#include <errno.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>

extern "C" void dsp_set_error_(const char* msg);

namespace {

constexpr uint64_t kMagic = 0x315247524e495244ull;  // "DRINGRG1"

struct Header {
    std::atomic<uint64_t> magic;      // written last by the creator: attachers wait for it
    uint64_t n_slots, slot_bytes;
    std::atomic<uint64_t> n_blocks;   // valid once finished != 0
    std::atomic<int32_t> finished;    // 1 = clean end of stream, < 0 = the producer failed (dsp_status)
    std::atomic<int32_t> aborted;     // a consumer failed: the producer stops waiting
    char message[256];                // the producer's error text
};
struct SlotHeader {
    std::atomic<uint64_t> published;  // seq + 1 of the block the slot holds
    std::atomic<uint64_t> released;   // seq + 1 of the last block a consumer has copied out of the slot
    uint64_t len, first_row, n_rows;
    uint64_t pad[3];
};
static_assert(sizeof(SlotHeader) == 64, "one cache line per slot header");

int ring_fail(int code, const char* fmt, const char* a = "", long long b = 0) {
    char buf[384];
    snprintf(buf, sizeof(buf), fmt, a, b);
    dsp_set_error_(buf);
    return code;
}

double now_s() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
// polling back-off: 50 us for the first ~2 ms of a wait (a slot about to be published), 1 ms after that (a block takes
// > 100 ms to inflate: idle ranks must not keep eight cores busy waking up)
void nap(int& spins) {
    timespec ts{0, (spins < 40 ? 50 : 1000) * 1000};
    ++spins;
    nanosleep(&ts, nullptr);
}

}  // namespace

struct dsp_shm_ring {
    std::string name;
    void* map = nullptr;
    size_t bytes = 0;
    Header* h = nullptr;
    SlotHeader* slots = nullptr;
    uint8_t* payload = nullptr;
    bool owner = false;
};

namespace {

dsp_shm_ring* map_ring(const char* name, int fd, size_t bytes, bool owner) {
    void* m = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    if (m == MAP_FAILED) return nullptr;
    dsp_shm_ring* r = new (std::nothrow) dsp_shm_ring();
    if (!r) { munmap(m, bytes); return nullptr; }
    r->name = name; r->map = m; r->bytes = bytes; r->owner = owner;
    r->h = (Header*)m;
    return r;
}
void bind_slots(dsp_shm_ring* r) {
    r->slots = (SlotHeader*)((uint8_t*)r->map + 4096);
    r->payload = (uint8_t*)r->map + 4096 + ((r->h->n_slots * sizeof(SlotHeader) + 4095) / 4096) * 4096;
}

}  // namespace

extern "C" {

// Producer side.  The memory is reserved up front (posix_fallocate): a /dev/shm that is too small says so here (DSP_ENOMEM;
// the caller falls back to per-rank inflation) instead of killing the process with SIGBUS at the first write.
dsp_shm_ring* dsp_shm_ring_create(const char* name, int32_t n_slots, uint64_t slot_bytes) {
    if (!name || name[0] != '/' || n_slots < 1 || slot_bytes < 1) { ring_fail(DSP_EINVAL, "dsp_shm_ring_create: bad arguments"); return nullptr; }
    slot_bytes = (slot_bytes + 4095) / 4096 * 4096;
    const size_t bytes = 4096 + (((size_t)n_slots * sizeof(SlotHeader) + 4095) / 4096) * 4096 + (size_t)n_slots * slot_bytes;
    shm_unlink(name);  // a leftover of a crashed run with the same name
    const int fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0) { ring_fail(DSP_EINVAL, "dsp_shm_ring_create: shm_open(%s) failed (errno %lld)", name, errno); return nullptr; }
    const int rc = posix_fallocate(fd, 0, (off_t)bytes);
    if (rc != 0) {
        close(fd); shm_unlink(name);
        ring_fail(DSP_ENOMEM, "dsp_shm_ring_create: cannot reserve the ring %s in shared memory (errno %lld)", name, rc);
        return nullptr;
    }
    dsp_shm_ring* r = map_ring(name, fd, bytes, true);
    close(fd);
    if (!r) { shm_unlink(name); ring_fail(DSP_ENOMEM, "dsp_shm_ring_create: mmap of %s failed", name); return nullptr; }
    memset(r->map, 0, 4096 + (size_t)n_slots * sizeof(SlotHeader));
    r->h->n_slots = (uint64_t)n_slots; r->h->slot_bytes = slot_bytes;
    bind_slots(r);
    r->h->magic.store(kMagic, std::memory_order_release);
    return r;
}

dsp_shm_ring* dsp_shm_ring_attach(const char* name, double timeout_s) {
    if (!name) { ring_fail(DSP_EINVAL, "dsp_shm_ring_attach: NULL name"); return nullptr; }
    const double t0 = now_s();
    int spins = 0;
    for (;;) {
        const int fd = shm_open(name, O_RDWR, 0600);
        if (fd >= 0) {
            struct stat sb;
            if (fstat(fd, &sb) == 0 && (size_t)sb.st_size >= 4096 + sizeof(SlotHeader)) {
                dsp_shm_ring* r = map_ring(name, fd, (size_t)sb.st_size, false);
                close(fd);
                if (!r) { ring_fail(DSP_ENOMEM, "dsp_shm_ring_attach: mmap of %s failed", name); return nullptr; }
                while (r->h->magic.load(std::memory_order_acquire) != kMagic) {
                    if (now_s() - t0 > timeout_s) { munmap(r->map, r->bytes); delete r; ring_fail(DSP_EINVAL, "dsp_shm_ring_attach: %s never initialised", name); return nullptr; }
                    nap(spins);
                }
                bind_slots(r);
                return r;
            }
            close(fd);
        }
        if (now_s() - t0 > timeout_s) { ring_fail(DSP_EINVAL, "dsp_shm_ring_attach: %s did not appear", name); return nullptr; }
        nap(spins);
    }
}

uint64_t dsp_shm_ring_slot_bytes(const dsp_shm_ring* r) { return r ? r->h->slot_bytes : 0; }

// producer: the payload of block `seq`, once the consumer of block seq - n_slots has let go of the slot
uint8_t* dsp_shm_ring_acquire(dsp_shm_ring* r, uint64_t seq, double timeout_s) {
    if (!r) { ring_fail(DSP_EINVAL, "dsp_shm_ring_acquire: NULL ring"); return nullptr; }
    const uint64_t n = r->h->n_slots;
    SlotHeader& s = r->slots[seq % n];
    const double t0 = now_s();
    int spins = 0;
    while (seq >= n && s.released.load(std::memory_order_acquire) != seq - n + 1) {
        if (r->h->aborted.load(std::memory_order_acquire)) { ring_fail(DSP_EINVAL, "dsp_shm_ring_acquire: a consumer of %s failed", r->name.c_str()); return nullptr; }
        if (now_s() - t0 > timeout_s) { ring_fail(DSP_EINVAL, "dsp_shm_ring_acquire: %s: slot not released in time", r->name.c_str()); return nullptr; }
        nap(spins);
    }
    return r->payload + (seq % n) * r->h->slot_bytes;
}

int32_t dsp_shm_ring_publish(dsp_shm_ring* r, uint64_t seq, uint64_t len, uint64_t first_row, uint64_t n_rows) {
    if (!r || len > r->h->slot_bytes) return ring_fail(DSP_EINVAL, "dsp_shm_ring_publish: bad arguments");
    SlotHeader& s = r->slots[seq % r->h->n_slots];
    s.len = len; s.first_row = first_row; s.n_rows = n_rows;
    s.published.store(seq + 1, std::memory_order_release);
    return 0;
}

// producer: end of the stream after n_blocks blocks (status 0), or failure (status < 0 with its message)
int32_t dsp_shm_ring_finish(dsp_shm_ring* r, uint64_t n_blocks, int32_t status, const char* message) {
    if (!r) return ring_fail(DSP_EINVAL, "dsp_shm_ring_finish: NULL ring");
    r->h->n_blocks.store(n_blocks, std::memory_order_relaxed);
    if (message) { strncpy(r->h->message, message, sizeof(r->h->message) - 1); r->h->message[sizeof(r->h->message) - 1] = 0; }
    r->h->finished.store(status < 0 ? status : 1, std::memory_order_release);
    return 0;
}

// consumer: 0 = block `seq` is there (*data points INTO the ring: copy it out, then release), 1 = the stream ended before
// block seq, < 0 = the producer failed (its message becomes dsp_last_error) or the wait timed out
int32_t dsp_shm_ring_wait(dsp_shm_ring* r, uint64_t seq, double timeout_s, const uint8_t** data, uint64_t* len,
                          uint64_t* first_row, uint64_t* n_rows) {
    if (!r || !data || !len) return ring_fail(DSP_EINVAL, "dsp_shm_ring_wait: NULL argument");
    SlotHeader& s = r->slots[seq % r->h->n_slots];
    const double t0 = now_s();
    int spins = 0;
    for (;;) {
        if (s.published.load(std::memory_order_acquire) == seq + 1) {
            *data = r->payload + (seq % r->h->n_slots) * r->h->slot_bytes;
            *len = s.len;
            if (first_row) *first_row = s.first_row;
            if (n_rows) *n_rows = s.n_rows;
            return 0;
        }
        const int32_t fin = r->h->finished.load(std::memory_order_acquire);
        if (fin < 0) { dsp_set_error_(r->h->message); return fin; }
        if (fin > 0 && seq >= r->h->n_blocks.load(std::memory_order_relaxed)) return 1;
        if (now_s() - t0 > timeout_s) return ring_fail(DSP_EINVAL, "dsp_shm_ring_wait: %s: block %lld not published in time", r->name.c_str(), (long long)seq);
        nap(spins);
    }
}

int32_t dsp_shm_ring_release(dsp_shm_ring* r, uint64_t seq) {
    if (!r) return ring_fail(DSP_EINVAL, "dsp_shm_ring_release: NULL ring");
    r->slots[seq % r->h->n_slots].released.store(seq + 1, std::memory_order_release);
    return 0;
}

// a failing consumer tells the producer to stop waiting for it
void dsp_shm_ring_abort(dsp_shm_ring* r) { if (r) r->h->aborted.store(1, std::memory_order_release); }

// remove the NAME once every consumer has attached: the memory lives on while it is mapped and goes away with the last
// process, however that process ends (a killed run leaves nothing behind in /dev/shm)
void dsp_shm_ring_unlink(dsp_shm_ring* r) {
    if (r && r->owner) { shm_unlink(r->name.c_str()); r->owner = false; }
}

void dsp_shm_ring_close(dsp_shm_ring* r, int32_t unlink_it) {
    if (!r) return;
    if (r->map) munmap(r->map, r->bytes);
    if (unlink_it && r->owner) shm_unlink(r->name.c_str());
    delete r;
}

}  // extern "C"
