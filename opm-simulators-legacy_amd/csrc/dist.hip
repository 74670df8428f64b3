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
// Two transports sit behind the two primitives (all-reduce, neighbour exchange): RCCL (loaded lazily with dlopen so that
// single-GPU use never touches it; EVERY return code is checked and becomes OPMGPU_ECOMM), or callbacks supplied by the caller
// (opmgpu_comm_init_transport: the host application's MPI communicator, or the shared-memory TEST transport that lives in
// tests/support -- no test scaffolding is compiled into this library).
#include <dlfcn.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>

#include "dist.hpp"

namespace opmgpu {

namespace {

// minimal RCCL (NCCL API) surface
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
enum { ncclSuccess = 0, ncclInProgress = 7 };
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
    int (*CommGetAsyncError)(ncclComm_t, int*) = nullptr;      // optional
    const char* (*GetErrorString)(int) = nullptr;              // optional
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
        *reinterpret_cast<void**>(&CommGetAsyncError) = dlsym(h, "ncclCommGetAsyncError");
        *reinterpret_cast<void**>(&GetErrorString) = dlsym(h, "ncclGetErrorString");
        return true;
    }
};
Rccl g_rccl;

// every RCCL call goes through here: a failed collective surfaces as OPMGPU_ECOMM instead of a hang or garbage halo values
void rccl_check(int rc, const char* what)
{
    if (rc == ncclSuccess) return;
    std::string msg = std::string("RCCL: ") + what + " failed";
    if (g_rccl.GetErrorString) { msg += ": "; msg += g_rccl.GetErrorString(rc); }
    else msg += " (code " + std::to_string(rc) + ")";
    throw HipError(OPMGPU_ECOMM, msg);
}

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

} // namespace

// the two primitives of the communicator
struct RcclComm::Transport {
    virtual ~Transport() {}
    virtual void allreduce(double* d, int n, bool is_max, hipStream_t s) = 0;
    // per neighbour q: send sbytes[q] from sb + soff[q], receive rbytes[q] into rb + roff[q]
    virtual void exchange(const std::vector<int32_t>& neigh, const char* sb, const std::vector<int64_t>& soff, const std::vector<int64_t>& sbytes,
                          char* rb, const std::vector<int64_t>& roff, const std::vector<int64_t>& rbytes, hipStream_t s) = 0;
    // the two at once (they are independent): one operation where the transport can fuse them
    virtual void allreduce_exchange(double* d, int n, const std::vector<int32_t>& neigh, const char* sb, const std::vector<int64_t>& soff,
                                    const std::vector<int64_t>& sbytes, char* rb, const std::vector<int64_t>& roff, const std::vector<int64_t>& rbytes, hipStream_t s)
    {
        allreduce(d, n, false, s);
        exchange(neigh, sb, soff, sbytes, rb, roff, rbytes, s);
    }
    virtual void check_async() {}
};

namespace {

struct RcclTransport : RcclComm::Transport {
    ncclComm_t comm = nullptr;
    ~RcclTransport() override { if (comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(comm); }
    void allreduce(double* d, int n, bool is_max, hipStream_t s) override
    {
        rccl_check(g_rccl.AllReduce(d, d, size_t(n), ncclFloat64, is_max ? ncclMax : ncclSum, comm, s), "ncclAllReduce");
    }
    void exchange(const std::vector<int32_t>& neigh, const char* sb, const std::vector<int64_t>& soff, const std::vector<int64_t>& sbytes,
                  char* rb, const std::vector<int64_t>& roff, const std::vector<int64_t>& rbytes, hipStream_t s) override
    {
        rccl_check(g_rccl.GroupStart(), "ncclGroupStart");
        int rc = ncclSuccess;
        for (size_t q = 0; q < neigh.size() && rc == ncclSuccess; ++q) {
            if (sbytes[q]) rc = g_rccl.Send(sb + soff[q], size_t(sbytes[q]), ncclInt8, neigh[q], comm, s);
            if (rc == ncclSuccess && rbytes[q]) rc = g_rccl.Recv(rb + roff[q], size_t(rbytes[q]), ncclInt8, neigh[q], comm, s);
        }
        const int rc_end = g_rccl.GroupEnd();          // always close the group, then report the first failure
        rccl_check(rc, "ncclSend / ncclRecv");
        rccl_check(rc_end, "ncclGroupEnd");
    }
    void allreduce_exchange(double* d, int n, const std::vector<int32_t>& neigh, const char* sb, const std::vector<int64_t>& soff,
                            const std::vector<int64_t>& sbytes, char* rb, const std::vector<int64_t>& roff, const std::vector<int64_t>& rbytes, hipStream_t s) override
    {
        // OPMGPU_RCCL_FUSED=1: one group -- RCCL launches the collective and the point-to-point transfers together (one launch latency instead
        // of two).  Mixing a collective with sends / receives in one group is NCCL API since 2.8, but it has run here with ONE rank only (no
        // multi-GPU node in this pool), so the DEFAULT issues the two operations back to back on the stream, the pattern every RCCL has
        // run for years: the first multi-rank run of this library should not depend on the less-trodden path
        static const bool fused = std::getenv("OPMGPU_RCCL_FUSED") && std::atoi(std::getenv("OPMGPU_RCCL_FUSED")) != 0;
        if (!fused) { allreduce(d, n, false, s); exchange(neigh, sb, soff, sbytes, rb, roff, rbytes, s); return; }
        rccl_check(g_rccl.GroupStart(), "ncclGroupStart");
        int rc = g_rccl.AllReduce(d, d, size_t(n), ncclFloat64, ncclSum, comm, s);
        for (size_t q = 0; q < neigh.size() && rc == ncclSuccess; ++q) {
            if (sbytes[q]) rc = g_rccl.Send(sb + soff[q], size_t(sbytes[q]), ncclInt8, neigh[q], comm, s);
            if (rc == ncclSuccess && rbytes[q]) rc = g_rccl.Recv(rb + roff[q], size_t(rbytes[q]), ncclInt8, neigh[q], comm, s);
        }
        const int rc_end = g_rccl.GroupEnd();
        rccl_check(rc, "ncclAllReduce / ncclSend / ncclRecv");
        rccl_check(rc_end, "ncclGroupEnd");
    }
    void check_async() override
    {
        if (!g_rccl.CommGetAsyncError) return;
        int err = ncclSuccess;
        rccl_check(g_rccl.CommGetAsyncError(comm, &err), "ncclCommGetAsyncError");
        if (err != ncclInProgress) rccl_check(err, "asynchronous communicator error");
    }
};

struct ExternalTransport : RcclComm::Transport {
    opmgpu_transport t;
    explicit ExternalTransport(const opmgpu_transport& t_) : t(t_) {}
    ~ExternalTransport() override { if (t.destroy) t.destroy(t.self); }
    void allreduce(double* d, int n, bool is_max, hipStream_t s) override
    {
        if (t.allreduce(t.self, d, n, is_max ? 1 : 0, s) != 0) throw HipError(OPMGPU_ECOMM, "external transport: all-reduce failed");
    }
    void exchange(const std::vector<int32_t>& neigh, const char* sb, const std::vector<int64_t>& soff, const std::vector<int64_t>& sbytes,
                  char* rb, const std::vector<int64_t>& roff, const std::vector<int64_t>& rbytes, hipStream_t s) override
    {
        if (t.exchange(t.self, int(neigh.size()), neigh.data(), sb, soff.data(), sbytes.data(), rb, roff.data(), rbytes.data(), s) != 0)
            throw HipError(OPMGPU_ECOMM, "external transport: neighbour exchange failed");
    }
    void allreduce_exchange(double* d, int n, const std::vector<int32_t>& neigh, const char* sb, const std::vector<int64_t>& soff,
                            const std::vector<int64_t>& sbytes, char* rb, const std::vector<int64_t>& roff, const std::vector<int64_t>& rbytes, hipStream_t s) override
    {
        if (!t.allreduce_exchange) { allreduce(d, n, false, s); exchange(neigh, sb, soff, sbytes, rb, roff, rbytes, s); return; }
        if (t.allreduce_exchange(t.self, d, n, int(neigh.size()), neigh.data(), sb, soff.data(), sbytes.data(), rb, roff.data(), rbytes.data(), s) != 0)
            throw HipError(OPMGPU_ECOMM, "external transport: fused all-reduce + neighbour exchange failed");
    }
};

} // namespace

RcclComm::RcclComm() {}
RcclComm::~RcclComm() { delete transport; }

int RcclComm::unique_id(uint8_t* id)
{
    if (!g_rccl.load()) return OPMGPU_ECOMM;
    ncclUniqueId u;
    if (g_rccl.GetUniqueId(&u) != ncclSuccess) return OPMGPU_ECOMM;
    static_assert(sizeof(ncclUniqueId) == OPMGPU_UNIQUE_ID_BYTES, "unique id size");
    std::memcpy(id, &u, sizeof(u));
    return OPMGPU_OK;
}

int RcclComm::init(int rank_, int nranks_, const uint8_t* id, const opmgpu_transport* ext, int n_owned_, int n_local_, int n_neigh,
                   const int32_t* neigh_rank_, const int32_t* send_ptr_, const int32_t* send_cells_, const int32_t* recv_ptr_, const int32_t* recv_cells_)
{
    rank = rank_; nranks = nranks_; n_owned = n_owned_; n_local = n_local_;
    neigh_rank.assign(neigh_rank_, neigh_rank_ + n_neigh);
    send_ptr.assign(send_ptr_, send_ptr_ + n_neigh + 1); recv_ptr.assign(recv_ptr_, recv_ptr_ + n_neigh + 1);
    send_cells.assign(send_cells_, send_cells_ + send_ptr[n_neigh]); recv_cells.assign(recv_cells_, recv_cells_ + recv_ptr[n_neigh]);
    for (int c : send_cells) if (c < 0 || c >= n_owned) return OPMGPU_EINVAL;
    for (int c : recv_cells) if (c < n_owned || c >= n_local) return OPMGPU_EINVAL;
    for (int q : neigh_rank) if (q < 0 || q >= nranks) return OPMGPU_EINVAL;
    if (ext) {
        if (!ext->allreduce || !ext->exchange) return OPMGPU_EINVAL;
        transport = new ExternalTransport(*ext);
        return OPMGPU_OK;
    }
    if (!id || !g_rccl.load()) return OPMGPU_ECOMM;
    ncclUniqueId u;
    std::memcpy(&u, id, sizeof(u));
    std::unique_ptr<RcclTransport> t(new RcclTransport());
    if (g_rccl.CommInitRank(&t->comm, nranks, u, rank) != ncclSuccess) return OPMGPU_ECOMM;
    transport = t.release();
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
        const int b = (user_m > 0 && m == user_m) ? user_blk[c] : int(std::min<long>(m - 1, long(c) * m / n_owned));
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

int RcclComm::set_coarse_blocks(int m, const int32_t* blk)
{
    if (m < 0 || m > 8 || (m > 0 && !blk)) return OPMGPU_EINVAL;
    for (int c = 0; c < n_owned && m > 0; ++c) if (blk[c] < 0 || blk[c] >= m) return OPMGPU_EINVAL;
    user_m = m;
    user_blk.assign(blk, blk + (m > 0 ? n_owned : 0));
    return OPMGPU_OK;
}

template <class S> void RcclComm::halo_t(S* v, hipStream_t s, double* red, int nred)
{
    const int ns = int(send_cells.size()), nr = int(recv_cells.size());
    S* sb = reinterpret_cast<S*>(sbuf.p); S* rb = reinterpret_cast<S*>(rbuf.p);
    if (ns) hipLaunchKernelGGL((k_halo_pack<S>), dim3(grid_for(ns)), dim3(kBlock), 0, s, ns, nbp, d_send_rows.p, v, sb);
    const size_t nq = neigh_rank.size();
    std::vector<int64_t> soff(nq), sby(nq), roff(nq), rby(nq);
    for (size_t q = 0; q < nq; ++q) {
        soff[q] = int64_t(3) * send_ptr[q] * sizeof(S); sby[q] = int64_t(3) * (send_ptr[q + 1] - send_ptr[q]) * sizeof(S);
        roff[q] = int64_t(3) * recv_ptr[q] * sizeof(S); rby[q] = int64_t(3) * (recv_ptr[q + 1] - recv_ptr[q]) * sizeof(S);
    }
    if (red) transport->allreduce_exchange(red, nred, neigh_rank, reinterpret_cast<const char*>(sb), soff, sby, reinterpret_cast<char*>(rb), roff, rby, s);
    else transport->exchange(neigh_rank, reinterpret_cast<const char*>(sb), soff, sby, reinterpret_cast<char*>(rb), roff, rby, s);
    if (nr) hipLaunchKernelGGL((k_halo_unpack<S>), dim3(grid_for(nr)), dim3(kBlock), 0, s, nr, nbp, d_recv_rows.p, rb, v);
}
void RcclComm::halo_exchange_f(float* v, hipStream_t s) { halo_t<float>(v, s); }
void RcclComm::halo_exchange_d(double* v, hipStream_t s) { halo_t<double>(v, s); }
void RcclComm::allreduce_sum_halo_f(double* d, int n, float* v, hipStream_t s) { halo_t<float>(v, s, d, n); }
void RcclComm::allreduce_sum_halo_d(double* d, int n, double* v, hipStream_t s) { halo_t<double>(v, s, d, n); }
void RcclComm::allreduce_sum(double* d, int n, hipStream_t s) { transport->allreduce(d, n, false, s); }
void RcclComm::allreduce_max(double* d, int n, hipStream_t s) { transport->allreduce(d, n, true, s); }
void RcclComm::check_async() { if (transport) transport->check_async(); }

} // namespace opmgpu
