// shm_transport.cpp -- TEST transport for the multi-rank tests (tests/test_gpu_dist_shm.py, bench.py rehearsal): the ranks are processes
// of ONE host (typically sharing one GPU) and exchange through a POSIX shared-memory segment with host staging and a spin barrier.
// It plugs into the product through the public transport hook (opmgpu_comm_init_transport, include/opmgpu.h) -- the same hook an
// integrator uses to run the halo exchange over the host application's MPI communicator -- so everything above the two primitives
// (send / receive lists, owner masks, pack / unpack kernels, the collectives' call sites in the solver) is the code the RCCL path runs.
// Test harness only: synchronous, slow, at most 8 ranks.  Built by tests/support/Makefile into tests/support/_build/libshmtransport.so.
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/opmgpu.h"

namespace {

constexpr int kMaxRanks = 8, kRedDoubles = 4096;
constexpr size_t kMailBytes = size_t(16) << 20;
struct Header {
    std::atomic<int> ready, count, gen;
    int nranks;
    double red[kMaxRanks][kRedDoubles];
    // per rank: the byte offset inside its mailbox of what it sends to each other rank, and the byte count
    int64_t soff[kMaxRanks][kMaxRanks], sbytes[kMaxRanks][kMaxRanks];
};
struct Shm {
    Header* hdr = nullptr;
    char* mail = nullptr;
    size_t bytes = 0;
    std::string name;
    int rank = 0, nranks = 1;
    std::vector<char> hs, hr;
    bool failed = false;

    bool open(const char* name_, int rank_, int nranks_)
    {
        if (nranks_ > kMaxRanks) return false;
        rank = rank_; nranks = nranks_; name = name_;
        bytes = sizeof(Header) + size_t(kMaxRanks) * kMailBytes;
        int fd = -1;
        const auto t0 = std::chrono::steady_clock::now();
        auto late = [&]() { return std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120); };
        if (rank == 0) {
            (void)shm_unlink(name.c_str());
            fd = shm_open(name.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
            if (fd < 0 || ftruncate(fd, off_t(bytes)) != 0) return false;
        } else {
            while ((fd = shm_open(name.c_str(), O_RDWR, 0600)) < 0) { if (late()) return false; usleep(1000); }
            while (lseek(fd, 0, SEEK_END) < off_t(bytes)) { if (late()) return false; usleep(1000); }
        }
        void* p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        close(fd);
        if (p == MAP_FAILED) return false;
        hdr = static_cast<Header*>(p);
        mail = static_cast<char*>(p) + sizeof(Header);
        if (rank == 0) { hdr->count.store(0); hdr->gen.store(0); hdr->nranks = nranks; hdr->ready.store(1); }
        else while (hdr->ready.load() != 1) { if (late()) return false; usleep(100); }
        return true;
    }
    bool barrier()
    {
        const int g = hdr->gen.load();
        if (hdr->count.fetch_add(1) + 1 == nranks) { hdr->count.store(0); hdr->gen.fetch_add(1); return true; }
        const auto t0 = std::chrono::steady_clock::now();
        for (long spins = 0; hdr->gen.load() == g; ++spins) {
            if ((spins & 0xffff) == 0xffff && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) {
                std::fprintf(stderr, "shm transport: a rank did not reach the barrier within 120 s\n");
                return false;
            }
            __builtin_ia32_pause();
        }
        return true;
    }
    long n_allreduce = 0, n_exchange = 0, n_fused = 0;
    ~Shm()
    {
        // OPMGPU_SHM_STATS=<path prefix>: the number of collective calls of this rank (tests count the all-reduces of an iteration)
        if (const char* e = std::getenv("OPMGPU_SHM_STATS")) {
            const std::string f = std::string(e) + "." + std::to_string(rank);
            if (FILE* fp = std::fopen(f.c_str(), "w")) { std::fprintf(fp, "%ld %ld %ld\n", n_allreduce, n_exchange, n_fused); std::fclose(fp); }
        }
        if (hdr) munmap(hdr, bytes);
        if (rank == 0 && !name.empty()) (void)shm_unlink(name.c_str());
    }
};

int shm_allreduce(void* self, double* d, int n, int is_max, void* stream)
{
    Shm& m = *static_cast<Shm*>(self);
    hipStream_t s = static_cast<hipStream_t>(stream);
    ++m.n_allreduce;
    if (n > kRedDoubles) return 1;
    std::vector<double> h(n);
    if (hipMemcpyAsync(h.data(), d, size_t(n) * sizeof(double), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return 1;
    std::memcpy(m.hdr->red[m.rank], h.data(), size_t(n) * sizeof(double));
    if (!m.barrier()) return 1;
    for (int i = 0; i < n; ++i) {            // rank order: the same bits on every rank
        double a = m.hdr->red[0][i];
        for (int r = 1; r < m.nranks; ++r) a = is_max ? std::max(a, m.hdr->red[r][i]) : a + m.hdr->red[r][i];
        h[i] = a;
    }
    if (!m.barrier()) return 1;
    if (hipMemcpyAsync(d, h.data(), size_t(n) * sizeof(double), hipMemcpyHostToDevice, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return 1;
    return 0;
}

int shm_exchange(void* self, int nq, const int32_t* neigh, const void* sbuf, const int64_t* soff, const int64_t* scount, void* rbuf,
                 const int64_t* roff, const int64_t* rcount, void* stream)
{
    Shm& m = *static_cast<Shm*>(self);
    hipStream_t s = static_cast<hipStream_t>(stream);
    ++m.n_exchange;
    int64_t stot = 0, rtot = 0;
    for (int q = 0; q < nq; ++q) { stot = std::max(stot, soff[q] + scount[q]); rtot = std::max(rtot, roff[q] + rcount[q]); }
    if (size_t(stot) > kMailBytes) return 1;
    m.hs.resize(std::max<size_t>(size_t(stot), 1)); m.hr.resize(std::max<size_t>(size_t(rtot), 1));
    if (stot && hipMemcpyAsync(m.hs.data(), sbuf, size_t(stot), hipMemcpyDeviceToHost, s) != hipSuccess) return 1;
    if (hipStreamSynchronize(s) != hipSuccess) return 1;
    std::memcpy(m.mail + size_t(m.rank) * kMailBytes, m.hs.data(), size_t(stot));
    for (int r = 0; r < m.nranks; ++r) { m.hdr->soff[m.rank][r] = 0; m.hdr->sbytes[m.rank][r] = 0; }
    for (int q = 0; q < nq; ++q) { m.hdr->soff[m.rank][neigh[q]] = soff[q]; m.hdr->sbytes[m.rank][neigh[q]] = scount[q]; }
    if (!m.barrier()) return 1;
    for (int q = 0; q < nq; ++q) {
        const int other = neigh[q];
        if (m.hdr->sbytes[other][m.rank] != rcount[q]) { std::fprintf(stderr, "shm transport: send and receive lists of two ranks do not pair up\n"); return 1; }
        std::memcpy(m.hr.data() + roff[q], m.mail + size_t(other) * kMailBytes + m.hdr->soff[other][m.rank], size_t(rcount[q]));
    }
    if (!m.barrier()) return 1;
    if (rtot && (hipMemcpyAsync(rbuf, m.hr.data(), size_t(rtot), hipMemcpyHostToDevice, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)) return 1;
    return 0;
}

// the fused operation (opmgpu_transport::allreduce_exchange): here simply both, counted as ONE operation -- what it is over RCCL (one group)
int shm_allreduce_exchange(void* self, double* d, int n, int nq, const int32_t* neigh, const void* sbuf, const int64_t* soff, const int64_t* scount, void* rbuf,
                           const int64_t* roff, const int64_t* rcount, void* stream)
{
    Shm& m = *static_cast<Shm*>(self);
    const int a = shm_allreduce(self, d, n, 0, stream);
    const int b = a ? a : shm_exchange(self, nq, neigh, sbuf, soff, scount, rbuf, roff, rcount, stream);
    --m.n_allreduce; if (!a) --m.n_exchange; ++m.n_fused;
    return b;
}

void shm_destroy(void* self) { delete static_cast<Shm*>(self); }

} // namespace

extern "C" int shm_transport_create(const char* segment_name, int rank, int nranks, opmgpu_transport* out)
{
    if (!segment_name || !out) return 1;
    Shm* m = new Shm();
    if (!m->open(segment_name, rank, nranks)) { delete m; return 1; }
    out->self = m; out->allreduce = &shm_allreduce; out->exchange = &shm_exchange; out->destroy = &shm_destroy;
    out->allreduce_exchange = std::getenv("OPMGPU_SHM_NO_FUSED") ? nullptr : &shm_allreduce_exchange;
    return 0;
}
