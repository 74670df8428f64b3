// dist.hip -- multi-GPU (RCCL over xGMI) entry points.  Filled in by the domain-decomposition work;
// until opmgpu_comm_init succeeds a context is single-GPU.
#include "common.hpp"

extern "C" {

int opmgpu_comm_unique_id(uint8_t* id)
{
    (void)id;
    return OPMGPU_ECOMM;
}

int opmgpu_comm_init(opmgpu_ctx* ctx, int rank, int nranks, const uint8_t* id, int32_t n_owned, int n_neigh, const int32_t* neigh_rank,
                     const int32_t* send_ptr, const int32_t* send_cells, const int32_t* recv_ptr, const int32_t* recv_cells)
{
    (void)ctx; (void)rank; (void)nranks; (void)id; (void)n_owned; (void)n_neigh; (void)neigh_rank; (void)send_ptr; (void)send_cells; (void)recv_ptr; (void)recv_cells;
    return OPMGPU_ECOMM;
}

} // extern "C"
