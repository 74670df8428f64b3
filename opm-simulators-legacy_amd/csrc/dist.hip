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

#include <cstring>
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

} // namespace

struct RcclComm::Impl { ncclComm_t comm = nullptr; };

RcclComm::RcclComm() : impl(new Impl()) {}
RcclComm::~RcclComm()
{
    if (impl->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(impl->comm);
    delete impl;
}

int RcclComm::unique_id(uint8_t* id)
{
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
    if (!g_rccl.load()) return OPMGPU_ECOMM;
    rank = rank_; nranks = nranks_; n_owned = n_owned_; n_local = n_local_;
    neigh_rank.assign(neigh_rank_, neigh_rank_ + n_neigh);
    send_ptr.assign(send_ptr_, send_ptr_ + n_neigh + 1); recv_ptr.assign(recv_ptr_, recv_ptr_ + n_neigh + 1);
    send_cells.assign(send_cells_, send_cells_ + send_ptr[n_neigh]); recv_cells.assign(recv_cells_, recv_cells_ + recv_ptr[n_neigh]);
    for (int c : send_cells) if (c < 0 || c >= n_owned) return OPMGPU_EINVAL;
    for (int c : recv_cells) if (c < n_owned || c >= n_local) return OPMGPU_EINVAL;
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

template <class S> void RcclComm::halo_t(S* v, hipStream_t s)
{
    const int ns = int(send_cells.size()), nr = int(recv_cells.size());
    S* sb = reinterpret_cast<S*>(sbuf.p); S* rb = reinterpret_cast<S*>(rbuf.p);
    if (ns) hipLaunchKernelGGL((k_halo_pack<S>), dim3(grid_for(ns)), dim3(kBlock), 0, s, ns, nbp, d_send_rows.p, v, sb);
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
void RcclComm::allreduce_sum(double* d, int n, hipStream_t s) { g_rccl.AllReduce(d, d, size_t(n), ncclFloat64, ncclSum, impl->comm, s); }
void RcclComm::allreduce_max(double* d, int n, hipStream_t s) { g_rccl.AllReduce(d, d, size_t(n), ncclFloat64, ncclMax, impl->comm, s); }

} // namespace opmgpu
