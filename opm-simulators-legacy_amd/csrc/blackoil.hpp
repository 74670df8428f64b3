// blackoil.hpp -- device-resident fully-implicit black-oil model (assembly, convergence, update).
//
// GPU replacement of BlackoilModelBase::{assemble, getConvergence, updateState}
// (opm/autodiff/BlackoilModelBase_impl.hpp:757-913, 1633-1857, 1147-1389) and of the Eigen
// AutoDiffBlock machinery underneath them: instead of O(100) sparse-matrix products per assembly
// the chain rule is applied per cell / per connection and the 3x3 blocks are written straight
// into the solver's SELL-64 matrix.
#ifndef OPMGPU_BLACKOIL_HPP
#define OPMGPU_BLACKOIL_HPP

#include "linsolver.hpp"

namespace opmgpu {

struct HystArgs;
struct TabX {
    const double *swof_dkrw, *swof_dkrow, *swof_dpcow, *sgof_dkrg, *sgof_dkrog, *sgof_dpcgo;
    const double *oil_drs, *oil_dinvb_sat, *oil_dinvbmu_sat, *oil_col_dinvb, *oil_col_dinvbmu;
    const double *gas_drvsat, *gas_dinvb_sat, *gas_dinvbmu_sat, *gas_col_dinvb, *gas_col_dinvbmu;
};
struct DevTables { opmgpu_tables t; TabX x; };

// per-cell VALUE planes a row's neighbours read (doubles, stride nbp): phase pressures p_w, p_g (p_o is the state's pressure), the
// densities, b * mobility, rs, rv.  No derivative plane exists: a row differentiates its connections with respect to its own
// variables only and writes the neighbour's off-diagonal block itself (k_assemble_rows).
enum { VP_PW = 0, VP_PG = 1, VP_RHO = 2 /* + phase */, VP_U = 5 /* + phase */, VP_RS = 8, VP_RV = 9, VP_COUNT = 10 };

constexpr int kRedPart = 32;        // d_red: [0, 32) results, per-workgroup partials behind them

class BlackoilDevice {
public:
    BlackoilDevice(hipStream_t s, LinSolver& ls, const opmgpu_grid* g, const opmgpu_tables* t, const opmgpu_params* prm);
    ~BlackoilDevice();

    int set_wells(int nw, const int32_t* connpos, const int32_t* cells);
    void set_state(const double* p, const double* sat, const double* rs, const double* rv, const int8_t* hc);
    void get_state(double* p, double* sat, double* rs, double* rv, int8_t* hc);
    void fluid_in_place(const int32_t* fipnum, int dims, double* fip_cells, double* values);      // computeFluidInPlace (:2263-2445)
    void region_state_sums(const int32_t* region, int nregions, double* sums);                   // RateConverter calcAverages (RateConverterLegacy.hpp:718-768)
    void voidage_coefficients(int n, const double* p, const double* rs, const double* rv, const int32_t* pvtreg, double* coeff);   // calcCoeff (:495-548)
    void assemble(double dt, bool initial);
    template <class MS> void assemble_kernels(double dt, bool initial, MS* A, bool props_only = false);
    bool assemble_single = false;      // precision of the coming solve (opmgpu_set_solve_precision)
    bool early_factor_pending = false;      // the assembly left the early ILU0 factorisation to the convergence check (LinSolver::factor_early_mode 2)
    void start_early_factor();
    int convergence(double dt, double* B3, double* CNV3, double* MB3, double* linf3, int* converged);
    void perf_props(double* out);
    void perf_pvt(const double* press, double* out);                 // host well model: PVT of the perforated cells at given pressures
    void perf_pvt_device(const double* press_dev, double* out_dev, const int32_t* gate);
    void average_b(double* B3);
    void binv_sums_device(double* out13_dev, double* scratch_dev);
    int add_well_terms(const double* resid_delta, int nblk, const int32_t* rc, const double* blocks);
    void add_well_rhs(const double* rhs_delta);
    void perf_dx(double* out);
    // right-hand side of the scaled system into the solver's b vector (precision S)
    template <class S> void build_rhs();
    template <class S> void store_dx();                 // solver x -> resident dx (double, internal planes)
    void dx_to_host(double* dx);                        // resident dx -> host, equation-major caller order
    void update_state(const double* dx_host, double relax);
    void stabilize_update(int relax_type, double omega);
    void save_state();
    void restore_state();
    double relative_change();
    void update_sat_oil_max();
    void set_sat_oil_max(const double* v);
    void get_sat_oil_max(double* v);
    int update_hysteresis();
    int set_hysteresis(const double* mdc_ow, const double* mdc_go);
    int get_hysteresis(double* mdc_ow, double* mdc_go, double* d_ow, double* d_go);
    void get_residual(double* r);
    // wells on the device (wells.hip)
    struct WellsDev;
    struct VfpDev;
    int set_device_wells(const opmgpu_wells* spec);
    int well_state_set(const double* bhp, const double* qs, const double* perf_press, const double* perf_rates);
    int well_state_get(double* bhp, double* qs, double* perf_press, double* perf_rates);
    int well_convergence(double* flux3, double* ctrl);
    // the well residuals ride on convergence()'s read-back (one host round trip instead of two per Newton iteration); valid until the
    // next assembly
    bool well_words_sources(const void*& e, int& ne, const void*& f) const;
    std::vector<uint32_t> well_words;
    bool well_words_valid = false;
    // decomposed runs: the wells' six convergence numbers, max-all-reduced with the cells' (h_red[13..18] after convergence())
    void well_conv_pack(double* d_out6);
    bool well_red_valid = false;
    bool has_device_wells() const;        // a device well model is attached (on this or, in a multi-rank run, on any rank)
    int set_vfp_tables(int n, const opmgpu_vfp_table* tabs);
    int well_controls_set(const int32_t* current, const double* thp);
    int well_controls_set_targets(const double* target, const double* distr);
    int well_controls_get(int32_t* current, double* thp, int32_t* pre_its, int32_t* pre_conv);
    void wells_recover();                                   // recoverVariable: well part of the increment from the resident dx
    bool device_wells = false;
    int n_owned_cells = 0;            // multi-GPU: caller cells [0, n_owned) are owned (set by attach_comm), else nc
    double time_assemble(int reps, int props_only);
    void attach_comm(CommBase* c, int n_owned);
    const Plan& plan() const { return ls.plan; }

    int nc = 0, nconn = 0;
    opmgpu_params prm;
    double last_dt = 0.0;
    bool has_state = false, has_dx = false, has_rhs_extra = false, has_saved = false;
    int nperf = 0;

private:
    void upload_tables(const opmgpu_tables* t);
    void perf_props_device();
    void wells_assemble(bool initial);
    // the part of wells_assemble() that reads only the state (perforation properties, updateWellControls): on Newton iterations after the
    // first it runs on a side stream next to the reservoir assembly (returns true; wells_assemble() then starts behind it)
    bool wells_prologue_async(bool initial);
    void wells_prologue();
    hipStream_t well_stream = nullptr;
    hipEvent_t ev_well[2] = { nullptr, nullptr };
    bool well_prologue_done = false;
    void wells_update(double relax, bool dx_from_host);
    void wells_stabilize(int sor, double omega);
    void wells_connection_pressures(const int32_t* gate);
    void wells_save();
    void wells_restore();
    void wells_rebind();
    void wells_free();
    void vfp_free();
    WellsDev* wd = nullptr;
    VfpDev* vfp = nullptr;
    std::vector<double> h_surface_density;
    void rebuild_structure();

    hipStream_t stream;
    LinSolver& ls;
    // host copies of the static inputs (caller numbering)
    std::vector<int32_t> h_conn, h_pvtnum, h_satnum, h_well_connpos, h_well_cells;
    std::vector<double> h_trans, h_pv, h_z, h_thpres;
    double gravity = 0.0, pvsum = 0.0, pvsum_global = 0.0;
    bool use_thpres = false;
    // ENDSCALE: per-cell scaled end points (caller numbering), unscaled points per saturation region
    bool use_eps = false;           // any saturation-function scaling: horizontal end points and / or vertical maxima
    bool has_endpoints = false, scalecrs = false, use_hyst = false, has_iendpoints = false;
    std::vector<double> h_eps[8], h_unscaled, h_eps_v[5], h_ieps[8];
    std::vector<int32_t> h_imbnum;
    DevArray<double> d_eps, d_eps_u0, d_somax, d_saved, d_ieps, d_ieps_u0, d_hist;     // d_hist: [mdc_ow | mdc_go | d_ow | d_go] planes
    DevArray<int32_t> d_imbnum;
    HystArgs hyst_args() const;
    void build_eps_planes(const std::vector<double>* ep8, bool have_points, bool imbibition, std::vector<double>& planes) const;
    std::vector<double> h_tabmax;
    DevArray<int8_t> d_saved_hc;
    const double* eps_planes() const { return use_eps ? d_eps.p : nullptr; }
    // device: tables
    opmgpu_tables dt_;                       // same struct, device pointers
    TabX dx_;                                // per-segment slopes of the 1-D tables (device pointers into the same blob)
    DevTables dto_;                          // tables + slopes with WORD OFFSETS into the blob in the pointer fields (resolve_tables)
    DevArray<double> d_tab;                  // all table arrays in one blob of 8-byte words (staged in LDS by the property kernels)
    int tab_words = 0;
    static constexpr int kTabLdsMaxBytes = 24 * 1024;     // 6 workgroups x 24 KiB fit the 160 KiB LDS of a CU: no occupancy lost
    int tab_lds_words() const { return tab_words * 8 <= kTabLdsMaxBytes ? tab_words : 0; }
    size_t tab_lds_bytes() const { return size_t(tab_lds_words()) * 8; }
    // device: static per-cell / per-connection (internal numbering for cells)
    bool dual_written = false;      // mixed precision: this assembly also wrote the float copy of its double Jacobian
    DevArray<double> d_pv, d_tr_e, d_thp_e;      // per SELL entry: +-transmissibility (sign = side, NaN = well fill), threshold pressure; d_zc: cell depths (g dz is formed in the kernel)
    DevArray<int32_t> d_pvtnum, d_satnum, d_perf_cells;
    // device: state (internal numbering)
    DevArray<double> d_p, d_sw, d_so, d_sg, d_rs, d_rv;
    DevArray<int8_t> d_hc;
    // device: work
    void launch_cell_values();
    DevArray<double> d_gather;      // decomposed runs: [ranks][19] table of getConvergence's sums and maxima (one all-reduce)
    DevArray<double> d_vals, d_accum0, d_R, d_bpart, d_zc, d_dx, d_dx_old, d_red, d_perf, d_rhs_extra;
    double* h_red = nullptr;
    std::vector<double> hbuf;
    std::vector<int8_t> hbuf8;
};

} // namespace opmgpu
#endif
