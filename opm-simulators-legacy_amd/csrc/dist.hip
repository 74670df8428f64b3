// dist.hip -- multi-GPU support: domain decomposition with a one-cell halo over RCCL (xGMI).
//
// Mirrors the reference's owner/overlap scheme (ParallelISTLInformation + OwnerOverlapCopyCommunication,
// ISTLSolver.hpp:286-298): every rank stores its owned cells plus one ghost layer; vectors are made
// consistent by copying owner values to the ghosts (copyOwnerToAll == halo_exchange), scalar products
// are sums over owned entries + all-reduce, the ILU0 is rank-local (block Jacobi, like
// ParallelOverlappingILU0 without the overlap update).  Messages are tiny (<= ~120 KB per neighbour,
// 8-16 B all-reduces): the path is latency bound, so everything is enqueued in-stream -- no host
// synchronisation between a kernel and the collective that consumes its output.
//
// RCCL is loaded lazily with dlopen so that single-GPU use never touches it.
#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>

#include "dist.hpp"

namespace opmgpu {

namespace {

// minimal RCCL (NCCL API) surface
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
enum { ncclSuccess = 0 };
enum { ncclInt8 = 0, ncclFloat32 = 7, ncclFloat64 = 8 };
enum { ncclSum = 0, ncclMax = 2 };

struct Rccl {
    void* h = nullptr;
    int (*GetUniqueId)(ncclUniqueId*) = nullptr;
    int (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*Send)(const void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    bool load()
    {
        if (h) return true;
        // Load the RCCL that sits next to the HIP runtime this process actually uses (PyTorch wheels bundle their own
        // libamdhip64 / libhsa-runtime64 / librccl; mixing one tree's RCCL with the other's HSA runtime opens a second
        // HSA instance that cannot see the GPU).
        std::string dir;
        Dl_info info;
        if (dladdr(reinterpret_cast<void*>(&hipGetDeviceCount), &info) && info.dli_fname) {
            dir = info.dli_fname;
            const size_t k = dir.rfind('/');
            dir = (k == std::string::npos) ? std::string() : dir.substr(0, k + 1);
        }
        std::vector<std::string> names;
        if (!dir.empty()) { names.push_back(dir + "librccl.so"); names.push_back(dir + "librccl.so.1"); }
        names.push_back("librccl.so"); names.push_back("librccl.so.1");
        names.push_back("/opt/rocm/lib/librccl.so"); names.push_back("/opt/rocm/lib/librccl.so.1");
        for (const std::string& n : names) { h = dlopen(n.c_str(), RTLD_NOW | RTLD_GLOBAL); if (h) break; }
        if (!h) return false;
#define L(sym) *reinterpret_cast<void**>(&sym) = dlsym(h, "nccl" #sym); if (!sym) return false;
        L(GetUniqueId) L(CommInitRank) L(CommDestroy) L(AllReduce) L(Send) L(Recv) L(GroupStart) L(GroupEnd)
#undef L
        return true;
    }
};
Rccl g_rccl;

template <class S>
__global__ __launch_bounds__(kBlock) void k_halo_pack(int n, int nbp, const int32_t* __restrict__ rows, const S* __restrict__ v, S* __restrict__ buf)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const int r = rows[i];
    buf[3 * long(i)] = v[r]; buf[3 * long(i) + 1] = v[nbp + r]; buf[3 * long(i) + 2] = v[2 * long(nbp) + r];
}
template <class S>
__global__ __launch_bounds__(kBlock) void k_halo_unpack(int n, int nbp, const int32_t* __restrict__ rows, const S* __restrict__ buf, S* __restrict__ v)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const int r = rows[i];
    v[r] = buf[3 * long(i)]; v[nbp + r] = buf[3 * long(i) + 1]; v[2 * long(nbp) + r] = buf[3 * long(i) + 2];
}


// ---- TEST transport (OPMGPU_COMM_TRANSPORT=shm): the ranks are processes of ONE host (typically sharing one GPU) and exchange through a
// POSIX shared-memory segment with host staging and a spin barrier.  Everything above the three primitives (all-reduce, halo
// exchange, communicator set-up) -- send / receive lists, owner masks, pack / unpack kernels, the collectives' call sites in the
// solver -- is the code the RCCL path runs, so real multi-rank runs can be tested where only one GPU is available
// (tests/test_gpu_dist_shm.py).  It is a test harness: synchronous, slow, at most 8 ranks.
constexpr int kShmMaxRanks = 8, kShmMaxNeigh = 16, kShmRedDoubles = 4096;
constexpr size_t kShmMailBytes = size_t(16) << 20;
struct ShmMeta { int n_neigh; int neigh_rank[kShmMaxNeigh]; int send_ptr[kShmMaxNeigh + 1]; };
struct ShmHeader {
    std::atomic<int> ready, count, gen;
    int nranks;
    ShmMeta meta[kShmMaxRanks];
    double red[kShmMaxRanks][kShmRedDoubles];
};
struct Shm {
    ShmHeader* hdr = nullptr;
    char* mail = nullptr;            // kShmMaxRanks mailboxes of kShmMailBytes
    size_t bytes = 0;
    std::string name;
    int rank = 0, nranks = 1;
    std::vector<char> hs, hr;        // host staging of the halo buffers
    static bool wanted() { const char* e = std::getenv("OPMGPU_COMM_TRANSPORT"); return e && std::string(e) == "shm"; }
    static std::string name_of(const uint8_t* id)
    {
        char buf[64];
        std::snprintf(buf, sizeof buf, "/opmgpu_%02x%02x%02x%02x%02x%02x%02x%02x", id[4], id[5], id[6], id[7], id[8], id[9], id[10], id[11]);
        return buf;
    }
    bool open(const uint8_t* id, int rank_, int nranks_)
    {
        if (nranks_ > kShmMaxRanks) return false;
        rank = rank_; nranks = nranks_; name = name_of(id);
        bytes = sizeof(ShmHeader) + size_t(kShmMaxRanks) * kShmMailBytes;
        int fd = -1;
        const auto t0 = std::chrono::steady_clock::now();
        if (rank == 0) {
            (void)shm_unlink(name.c_str());
            fd = shm_open(name.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
            if (fd < 0 || ftruncate(fd, off_t(bytes)) != 0) return false;
        } else {
            while ((fd = shm_open(name.c_str(), O_RDWR, 0600)) < 0) {
                if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) return false;
                usleep(1000);
            }
            while (lseek(fd, 0, SEEK_END) < off_t(bytes)) {          // until rank 0 has sized it
                if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) return false;
                usleep(1000);
            }
        }
        void* p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        close(fd);
        if (p == MAP_FAILED) return false;
        hdr = static_cast<ShmHeader*>(p);
        mail = static_cast<char*>(p) + sizeof(ShmHeader);
        if (rank == 0) { hdr->count.store(0); hdr->gen.store(0); hdr->nranks = nranks; hdr->ready.store(1); }
        else while (hdr->ready.load() != 1) { if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) return false; usleep(100); }
        return true;
    }
    void barrier()
    {
        const int g = hdr->gen.load();
        if (hdr->count.fetch_add(1) + 1 == nranks) { hdr->count.store(0); hdr->gen.fetch_add(1); return; }
        const auto t0 = std::chrono::steady_clock::now();
        for (long spins = 0; hdr->gen.load() == g; ++spins) {
            if ((spins & 0xffff) == 0xffff && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120))
                throw HipError(OPMGPU_ECOMM, "shm transport: a rank did not reach the barrier within 120 s");
            __builtin_ia32_pause();
        }
    }
    ~Shm()
    {
        if (hdr) munmap(hdr, bytes);
        if (rank == 0 && !name.empty()) (void)shm_unlink(name.c_str());
    }
};

} // namespace

struct RcclComm::Impl { ncclComm_t comm = nullptr; std::unique_ptr<Shm> shm; };

RcclComm::RcclComm() : impl(new Impl()) {}
RcclComm::~RcclComm()
{
    if (impl->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(impl->comm);
    delete impl;
}

int RcclComm::unique_id(uint8_t* id)
{
    if (Shm::wanted()) {          // test transport: a random segment name instead of an RCCL id
        std::memset(id, 0, OPMGPU_UNIQUE_ID_BYTES);
        std::memcpy(id, "SHM1", 4);
        FILE* f = std::fopen("/dev/urandom", "rb");
        const bool ok = f && std::fread(id + 4, 1, 16, f) == 16;
        if (f) std::fclose(f);
        return ok ? OPMGPU_OK : OPMGPU_ECOMM;
    }
    if (!g_rccl.load()) return OPMGPU_ECOMM;
    ncclUniqueId u;
    if (g_rccl.GetUniqueId(&u) != ncclSuccess) return OPMGPU_ECOMM;
    static_assert(sizeof(ncclUniqueId) == OPMGPU_UNIQUE_ID_BYTES, "unique id size");
    std::memcpy(id, &u, sizeof(u));
    return OPMGPU_OK;
}

int RcclComm::init(int rank_, int nranks_, const uint8_t* id, int n_owned_, int n_local_, int n_neigh, const int32_t* neigh_rank_,
                   const int32_t* send_ptr_, const int32_t* send_cells_, const int32_t* recv_ptr_, const int32_t* recv_cells_)
{
    rank = rank_; nranks = nranks_; n_owned = n_owned_; n_local = n_local_;
    neigh_rank.assign(neigh_rank_, neigh_rank_ + n_neigh);
    send_ptr.assign(send_ptr_, send_ptr_ + n_neigh + 1); recv_ptr.assign(recv_ptr_, recv_ptr_ + n_neigh + 1);
    send_cells.assign(send_cells_, send_cells_ + send_ptr[n_neigh]); recv_cells.assign(recv_cells_, recv_cells_ + recv_ptr[n_neigh]);
    for (int c : send_cells) if (c < 0 || c >= n_owned) return OPMGPU_EINVAL;
    for (int c : recv_cells) if (c < n_owned || c >= n_local) return OPMGPU_EINVAL;
    if (std::memcmp(id, "SHM1", 4) == 0) {
        if (n_neigh > kShmMaxNeigh) return OPMGPU_EINVAL;
        impl->shm.reset(new Shm());
        if (!impl->shm->open(id, rank, nranks)) { impl->shm.reset(); return OPMGPU_ECOMM; }
        ShmMeta& m = impl->shm->hdr->meta[rank];
        m.n_neigh = n_neigh;
        for (int q = 0; q < n_neigh; ++q) m.neigh_rank[q] = neigh_rank[q];
        for (int q = 0; q <= n_neigh; ++q) m.send_ptr[q] = send_ptr[q];
        impl->shm->barrier();
        return OPMGPU_OK;
    }
    if (!g_rccl.load()) return OPMGPU_ECOMM;
    ncclUniqueId u;
    std::memcpy(&u, id, sizeof(u));
    if (g_rccl.CommInitRank(&impl->comm, nranks, u, rank) != ncclSuccess) return OPMGPU_ECOMM;
    return OPMGPU_OK;
}

void RcclComm::rebuild(const Plan& P, hipStream_t s)
{
    nbp = P.nbp;
    std::vector<int32_t> sr(std::max<size_t>(send_cells.size(), 1), 0), rr(std::max<size_t>(recv_cells.size(), 1), 0);
    for (size_t i = 0; i < send_cells.size(); ++i) sr[i] = P.pos[send_cells[i]];
    for (size_t i = 0; i < recv_cells.size(); ++i) rr[i] = P.pos[recv_cells[i]];
    d_send_rows.upload(sr, s); d_recv_rows.upload(rr, s);
    std::vector<int8_t> m(P.nbp, 0);
    for (int c = 0; c < n_owned; ++c) m[P.pos[c]] = 1;
    d_mask.upload(m, s);
    sbuf.alloc(std::max<size_t>(send_cells.size(), 1) * 3); rbuf.alloc(std::max<size_t>(recv_cells.size(), 1) * 3);
    OPMGPU_HIP(hipStreamSynchronize(s));
}

void RcclComm::subdomain_of_rows(const Plan& P, std::vector<int32_t>& sub) const
{
    sub.assign(P.nbp, rank);
    for (size_t q = 0; q < neigh_rank.size(); ++q)
        for (int k = recv_ptr[q]; k < recv_ptr[q + 1]; ++k) sub[P.pos[recv_cells[k]]] = neigh_rank[q];
}

void RcclComm::coarse_blocks_of_rows(const Plan& P, int m, hipStream_t s, std::vector<int32_t>& sub, std::vector<int8_t>& blk)
{
    sub.assign(P.nbp, rank * m); blk.assign(P.nbp, int8_t(-1));
    std::vector<double> h(size_t(3) * P.nbp, 0.0);
    for (int c = 0; c < n_owned; ++c) {
        const int b = int(std::min<long>(m - 1, long(c) * m / n_owned));
        sub[P.pos[c]] = rank * m + b; blk[P.pos[c]] = int8_t(b);
        h[P.pos[c]] = double(rank * m + b);
    }
    // the ghosts' values come from their owners: one halo exchange of a plane vector carrying the ids (exact in a double)
    DevArray<double> tmp; tmp.alloc(size_t(3) * P.nbp);
    tmp.upload(h, s);
    halo_exchange_d(tmp.p, s);
    tmp.download(h.data(), h.size(), s);
    OPMGPU_HIP(hipStreamSynchronize(s));
    for (int c = n_owned; c < n_local; ++c) sub[P.pos[c]] = int32_t(h[P.pos[c]] + 0.5);
}

template <class S> void RcclComm::halo_t(S* v, hipStream_t s)
{
    const int ns = int(send_cells.size()), nr = int(recv_cells.size());
    S* sb = reinterpret_cast<S*>(sbuf.p); S* rb = reinterpret_cast<S*>(rbuf.p);
    if (ns) hipLaunchKernelGGL((k_halo_pack<S>), dim3(grid_for(ns)), dim3(kBlock), 0, s, ns, nbp, d_send_rows.p, v, sb);
    if (impl->shm) {
        Shm& m = *impl->shm;
        const size_t sbytes = size_t(3) * ns * sizeof(S), rbytes = size_t(3) * nr * sizeof(S);
        if (sbytes > kShmMailBytes) throw HipError(OPMGPU_ECOMM, "shm transport: halo larger than the mailbox");
        m.hs.resize(std::max<size_t>(sbytes, 1)); m.hr.resize(std::max<size_t>(rbytes, 1));
        if (ns) OPMGPU_HIP(hipMemcpyAsync(m.hs.data(), sb, sbytes, hipMemcpyDeviceToHost, s));
        OPMGPU_HIP(hipStreamSynchronize(s));
        std::memcpy(m.mail + size_t(rank) * kShmMailBytes, m.hs.data(), sbytes);
        m.barrier();
        for (size_t q = 0; q < neigh_rank.size(); ++q) {
            const int other = neigh_rank[q], cr = recv_ptr[q + 1] - recv_ptr[q];
            const ShmMeta& om = m.hdr->meta[other];
            int k = -1;
            for (int t = 0; t < om.n_neigh; ++t) if (om.neigh_rank[t] == rank) k = t;
            if (k < 0 || om.send_ptr[k + 1] - om.send_ptr[k] != cr) throw HipError(OPMGPU_ECOMM, "shm transport: send and receive lists of two ranks do not pair up");
            std::memcpy(m.hr.data() + size_t(3) * recv_ptr[q] * sizeof(S), m.mail + size_t(other) * kShmMailBytes + size_t(3) * om.send_ptr[k] * sizeof(S),
                        size_t(3) * cr * sizeof(S));
        }
        m.barrier();
        if (nr) { OPMGPU_HIP(hipMemcpyAsync(rb, m.hr.data(), rbytes, hipMemcpyHostToDevice, s)); OPMGPU_HIP(hipStreamSynchronize(s)); }
        if (nr) hipLaunchKernelGGL((k_halo_unpack<S>), dim3(grid_for(nr)), dim3(kBlock), 0, s, nr, nbp, d_recv_rows.p, rb, v);
        return;
    }
    const int type = sizeof(S) == 4 ? ncclFloat32 : ncclFloat64;
    g_rccl.GroupStart();
    for (size_t q = 0; q < neigh_rank.size(); ++q) {
        const int cs = send_ptr[q + 1] - send_ptr[q], cr = recv_ptr[q + 1] - recv_ptr[q];
        if (cs) g_rccl.Send(sb + 3 * size_t(send_ptr[q]), size_t(3) * cs, type, neigh_rank[q], impl->comm, s);
        if (cr) g_rccl.Recv(rb + 3 * size_t(recv_ptr[q]), size_t(3) * cr, type, neigh_rank[q], impl->comm, s);
    }
    g_rccl.GroupEnd();
    if (nr) hipLaunchKernelGGL((k_halo_unpack<S>), dim3(grid_for(nr)), dim3(kBlock), 0, s, nr, nbp, d_recv_rows.p, rb, v);
}
void RcclComm::halo_exchange_f(float* v, hipStream_t s) { halo_t<float>(v, s); }
void RcclComm::halo_exchange_d(double* v, hipStream_t s) { halo_t<double>(v, s); }
void RcclComm::shm_allreduce(double* d, int n, bool is_max, hipStream_t s)
{
    Shm& m = *impl->shm;
    if (n > kShmRedDoubles) throw HipError(OPMGPU_ECOMM, "shm transport: all-reduce too long");
    std::vector<double> h(n);
    OPMGPU_HIP(hipMemcpyAsync(h.data(), d, size_t(n) * sizeof(double), hipMemcpyDeviceToHost, s));
    OPMGPU_HIP(hipStreamSynchronize(s));
    std::memcpy(m.hdr->red[rank], h.data(), size_t(n) * sizeof(double));
    m.barrier();
    for (int i = 0; i < n; ++i) {            // rank order: the same bits on every rank
        double a = m.hdr->red[0][i];
        for (int r = 1; r < nranks; ++r) a = is_max ? std::max(a, m.hdr->red[r][i]) : a + m.hdr->red[r][i];
        h[i] = a;
    }
    m.barrier();
    OPMGPU_HIP(hipMemcpyAsync(d, h.data(), size_t(n) * sizeof(double), hipMemcpyHostToDevice, s));
    OPMGPU_HIP(hipStreamSynchronize(s));
}
void RcclComm::allreduce_sum(double* d, int n, hipStream_t s)
{
    if (impl->shm) { shm_allreduce(d, n, false, s); return; }
    g_rccl.AllReduce(d, d, size_t(n), ncclFloat64, ncclSum, impl->comm, s);
}
void RcclComm::allreduce_max(double* d, int n, hipStream_t s)
{
    if (impl->shm) { shm_allreduce(d, n, true, s); return; }
    g_rccl.AllReduce(d, d, size_t(n), ncclFloat64, ncclMax, impl->comm, s);
}

} // namespace opmgpu
