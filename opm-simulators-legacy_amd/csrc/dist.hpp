// dist.hpp -- RCCL communicator behind the CommBase hooks of the solver (see dist.hip).
#ifndef OPMGPU_DIST_HPP
#define OPMGPU_DIST_HPP

#include "linsolver.hpp"

namespace opmgpu {

class RcclComm : public CommBase {
public:
    RcclComm();
    ~RcclComm() override;
    static int unique_id(uint8_t* id);
    // cells in rank-local caller numbering: [0, n_owned) owned, [n_owned, n_local) ghosts
    // id: RCCL unique id (transport == nullptr), or an external transport (opmgpu_comm_init_transport)
    int init(int rank, int nranks, const uint8_t* id, const opmgpu_transport* transport, int n_owned, int n_local, int n_neigh, const int32_t* neigh_rank,
             const int32_t* send_ptr, const int32_t* send_cells, const int32_t* recv_ptr, const int32_t* recv_cells);
    void check_async() override;                         // asynchronous communicator errors (ncclCommGetAsyncError) -> OPMGPU_ECOMM
    struct Transport;
    void rebuild(const Plan& P, hipStream_t s);          // internal row lists + owner mask for the current plan
    void halo_exchange_f(float* v, hipStream_t s) override;
    void halo_exchange_d(double* v, hipStream_t s) override;
    void allreduce_sum(double* dbuf, int n, hipStream_t s) override;
    void allreduce_max(double* dbuf, int n, hipStream_t s) override;
    void allreduce_sum_halo_f(double* dbuf, int n, float* v, hipStream_t s) override;
    void allreduce_sum_halo_d(double* dbuf, int n, double* v, hipStream_t s) override;
    const int8_t* owner_mask() const override { return d_mask.p; }
    int my_rank() const override { return rank; }
    int num_ranks() const override { return nranks; }
    void subdomain_of_rows(const Plan& P, std::vector<int32_t>& sub) const override;
    void coarse_blocks_of_rows(const Plan& P, int m, hipStream_t s, std::vector<int32_t>& sub, std::vector<int8_t>& blk) override;

    int user_coarse_blocks() const override { return user_m; }
    int set_coarse_blocks(int m, const int32_t* block_of_owned_cell);      // 0 <= block < m per owned cell (local caller numbering); m <= 8
    std::vector<int32_t> user_blk;
    int user_m = 0;
    int rank = 0, nranks = 1, n_owned = 0, n_local = 0, nbp = 0;
    double pvsum_global = 0.0;

private:
    template <class S> void halo_t(S* v, hipStream_t s, double* red = nullptr, int nred = 0);
    Transport* transport = nullptr;
    std::vector<int32_t> neigh_rank, send_ptr, recv_ptr, send_cells, recv_cells;
    DevArray<int32_t> d_send_rows, d_recv_rows;
    DevArray<int8_t> d_mask;
    DevArray<double> sbuf, rbuf;
};

} // namespace opmgpu
#endif
