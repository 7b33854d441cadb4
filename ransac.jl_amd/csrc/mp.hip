// mp.hip -- rh_mp_*: the node-local exchange behind rh_ransac_mp (driver_internal.h has the segment's description)
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include "driver_internal.h"

namespace rhdrv {


constexpr uint64_t RH_MP_MAGIC = 0x52484d5032303236ULL;   // "RHMP2026"
constexpr int RH_MP_MAX_WORLD = 64;
// magic, world, then from byte 256 on one 64-byte line per (rank, parity) flag -- for the largest group rh_mp_open accepts
// (with a 4096-byte header the flags of ranks >= 30 lay inside rank 0's first slot)
constexpr size_t RH_MP_HDR = (256 + 64 * 2 * (size_t)RH_MP_MAX_WORLD + 4095) / 4096 * 4096;
static_assert(256 + 64 * 2 * (size_t)RH_MP_MAX_WORLD <= RH_MP_HDR, "the flags must not reach into the slots");

inline volatile uint64_t *mp_flag(rh_mp *m, int rank, int parity) { return (volatile uint64_t *)(m->base + 256 + 64 * (size_t)(rank * 2 + parity)); }
inline char *mp_slot(rh_mp *m, int rank, int parity) { return m->base + RH_MP_HDR + (size_t)(rank * 2 + parity) * (size_t)m->slot_bytes; }

// every rank publishes `bytes` of payload; afterwards payload r of every rank r can be read with mp_slot(m, r, parity)
// until the exchange after the next one.  Returns the parity used.
int mp_exchange(rh_mp *m, const void *payload, int64_t bytes, int *parity_out)
{
    if (bytes > m->slot_bytes) {
        rh_set_error("rh_ransac_mp: a window's candidate list (%lld bytes) does not fit the exchange slot (%lld)", (long long)bytes,
                     (long long)m->slot_bytes);
        return RH_E_CAPACITY;
    }
    m->seq++;
    const int par = (int)(m->seq & 1);
    memcpy(mp_slot(m, m->rank, par), payload, (size_t)bytes);
    __atomic_store_n((uint64_t *)mp_flag(m, m->rank, par), m->seq, __ATOMIC_RELEASE);
    const double t0 = now_s();
    for (int r = 0; r < m->world; r++) {
        uint64_t spins = 0;
        while (__atomic_load_n((uint64_t *)mp_flag(m, r, par), __ATOMIC_ACQUIRE) < m->seq) {
            if (++spins > 2000) {
                sched_yield();
                if ((spins & 1023) == 0 && now_s() - t0 > 60.0) {
                    rh_set_error("rh_ransac_mp: rank %d did not reach exchange %llu within 60 s (rank %d waited)", r,
                                 (unsigned long long)m->seq, m->rank);
                    return RH_E_INTERNAL;
                }
            }
        }
    }
    *parity_out = par;
    return RH_OK;
}

// the same for payloads of any size (and different sizes per rank): the payload travels in pieces of the slot size;
// every piece carries the rank's total, so after the first round all ranks agree on the number of rounds.
// recv[r] = rank r's payload.
int mp_exchange_any(rh_mp *m, const void *payload, int64_t bytes, std::vector<std::vector<char>> &recv)
{
    const int64_t cap = m->slot_bytes - 16;
    recv.assign((size_t)m->world, std::vector<char>());
    std::vector<char> piece((size_t)m->slot_bytes);
    int64_t rounds = 1;
    for (int64_t r = 0; r < rounds; r++) {
        const int64_t off = std::min(bytes, r * cap), len = std::min(cap, bytes - off);
        memcpy(piece.data(), &bytes, 8);
        memcpy(piece.data() + 8, &len, 8);
        if (len > 0) memcpy(piece.data() + 16, (const char *)payload + off, (size_t)len);
        int par = 0;
        RH_TRY(mp_exchange(m, piece.data(), 16 + len, &par));
        for (int k = 0; k < m->world; k++) {
            int64_t tot = 0, ln = 0;
            memcpy(&tot, mp_slot(m, k, par), 8);
            memcpy(&ln, mp_slot(m, k, par) + 8, 8);
            if (tot < 0 || ln < 0 || ln > cap) { rh_set_error("rh_ransac_mp: corrupt exchange header from rank %d", k); return RH_E_INTERNAL; }
            if (r == 0) {
                recv[(size_t)k].reserve((size_t)tot);
                rounds = std::max(rounds, (tot + cap - 1) / cap);
            }
            recv[(size_t)k].insert(recv[(size_t)k].end(), mp_slot(m, k, par) + 16, mp_slot(m, k, par) + 16 + ln);
        }
    }
    return RH_OK;
}

}  // namespace rhdrv

using namespace rhdrv;

extern "C" int rh_mp_open(const char *name, int32_t rank, int32_t world, int64_t slot_bytes, rh_mp **out)
{
    if (!name || !out || world < 1 || rank < 0 || rank >= world || world > RH_MP_MAX_WORLD || strlen(name) >= 120) {
        rh_set_error("rh_mp_open: bad arguments");
        return RH_E_INVALID;
    }
    if (slot_bytes <= 0) slot_bytes = (int64_t)1 << 20;
    slot_bytes = (slot_bytes + 4095) / 4096 * 4096;
    rh_mp *m = new rh_mp;
    m->rank = rank; m->world = world; m->slot_bytes = slot_bytes;
    m->map_bytes = RH_MP_HDR + (size_t)world * 2 * (size_t)slot_bytes;
    snprintf(m->name, sizeof m->name, "%s", name);
    int fd = -1;
    const double t0 = now_s();
    if (rank == 0) {
        (void)shm_unlink(name);
        fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd < 0 || ftruncate(fd, (off_t)m->map_bytes) != 0) {
            rh_set_error("rh_mp_open: cannot create shared memory %s (%lld bytes)", name, (long long)m->map_bytes);
            if (fd >= 0) close(fd);
            delete m;
            return RH_E_NOMEM;
        }
    } else {
        for (;;) {   // wait for rank 0 to create and size the segment
            fd = shm_open(name, O_RDWR, 0600);
            struct stat st;
            if (fd >= 0 && fstat(fd, &st) == 0 && (size_t)st.st_size >= m->map_bytes) break;
            if (fd >= 0) { close(fd); fd = -1; }
            if (now_s() - t0 > 60.0) { rh_set_error("rh_mp_open: rank 0 did not create %s within 60 s", name); delete m; return RH_E_INTERNAL; }
            usleep(1000);
        }
    }
    void *mem = mmap(nullptr, m->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (mem == MAP_FAILED) { rh_set_error("rh_mp_open: mmap of %s failed", name); delete m; return RH_E_NOMEM; }
    m->base = (char *)mem;
    if (rank == 0) {   // a fresh segment is zero-filled: flags start at 0; publish the header last
        ((volatile int32_t *)(m->base + 8))[0] = world;
        __atomic_store_n((uint64_t *)m->base, RH_MP_MAGIC, __ATOMIC_RELEASE);
    } else {
        while (__atomic_load_n((uint64_t *)m->base, __ATOMIC_ACQUIRE) != RH_MP_MAGIC) {
            if (now_s() - t0 > 60.0) { rh_set_error("rh_mp_open: %s was never initialised", name); munmap(mem, m->map_bytes); delete m; return RH_E_INTERNAL; }
            usleep(200);
        }
        if (((volatile int32_t *)(m->base + 8))[0] != world) {
            rh_set_error("rh_mp_open: %s was created for %d ranks, this is rank %d of %d", name, ((volatile int32_t *)(m->base + 8))[0], rank, world);
            munmap(mem, m->map_bytes);
            delete m;
            return RH_E_INVALID;
        }
    }
    // everybody has mapped the segment once this first exchange returns: the name can go (no leak if a rank dies later)
    int par = 0;
    const int32_t hello = rank;
    int rc = mp_exchange(m, &hello, sizeof hello, &par);
    if (rc != RH_OK) { munmap(mem, m->map_bytes); delete m; return rc; }
    if (rank == 0) (void)shm_unlink(name);
    *out = m;
    return RH_OK;
}

// the exchange on its own (host memory only, no GPU involved): every rank contributes `bytes` bytes (the same number on
// every rank), out receives world x bytes in rank order
extern "C" int rh_mp_allgather(rh_mp *m, const void *payload, int64_t bytes, void *out)
{
    if (!m || bytes < 0 || (bytes > 0 && (!payload || !out))) { rh_set_error("rh_mp_allgather: bad arguments"); return RH_E_INVALID; }
    std::vector<std::vector<char>> recv;   // (payloads beyond the slot size travel in pieces)
    RH_TRY(mp_exchange_any(m, payload, bytes, recv));
    for (int r = 0; r < m->world; r++) {
        if ((int64_t)recv[(size_t)r].size() != bytes) { rh_set_error("rh_mp_allgather: rank %d sent %lld bytes, expected %lld", r, (long long)recv[(size_t)r].size(), (long long)bytes); return RH_E_INVALID; }
        if (bytes > 0) memcpy((char *)out + (size_t)r * (size_t)bytes, recv[(size_t)r].data(), (size_t)bytes);
    }
    return RH_OK;
}

extern "C" int rh_mp_close(rh_mp *m)
{
    if (!m) return RH_OK;
    if (m->base) munmap(m->base, m->map_bytes);
    delete m;
    return RH_OK;
}
