// host_check -- the C++ host mirror (opmgpu.hpp) driven the way flow_legacy drives BlackoilModel / NonlinearSolver:
// a hand-authored dead-oil / dry-gas 6x6x3 deck, one time step through NonlinearSolverGpu::step (assemble ->
// getConvergence -> solveJacobianSystem -> stabilise -> updateState on the device), state downloaded at the end.
// Without a GPU it only exercises the "no device, no CPU fallback" error path (exit code 0 either way when behaving).
#include <cmath>
#include <cstdio>
#include <vector>

#include "opmgpu.hpp"

namespace {

struct Deck {
    // fluids, SI
    std::vector<double> dens{ 1000.0, 850.0, 1.0 }, pvtw{ 200e5, 1.02, 4.5e-10, 0.5e-3, 0.0 };
    std::vector<int32_t> oil_node_ptr{ 0, 2 }, oil_col_ptr{ 0, 1, 2 }, gas_node_ptr{ 0, 2 }, gas_col_ptr{ 0, 1, 2 };
    std::vector<double> oil_rs{ 0.0, 0.0 }, oil_p{ 1e5, 400e5 }, oil_invb{ 1.0, 1.05 }, oil_invbmu{ 1.0 / 1e-3, 1.05 / 1e-3 };
    std::vector<double> gas_p{ 1e5, 400e5 }, gas_rv{ 0.0, 0.0 }, gas_invb{ 1.0, 300.0 }, gas_invbmu{ 1.0 / 2e-5, 300.0 / 2e-5 };
    std::vector<int32_t> swof_ptr{ 0, 4 }, sgof_ptr{ 0, 4 };
    std::vector<double> sw{ 0.2, 0.5, 0.8, 1.0 }, krw{ 0, 0.2, 0.6, 1.0 }, krow{ 1.0, 0.3, 0.0, 0.0 }, pcow{ 0, 0, 0, 0 };
    std::vector<double> sg{ 0.0, 0.1, 0.5, 0.8 }, krg{ 0, 0, 0.4, 0.9 }, krog{ 1.0, 0.7, 0.1, 0.0 }, pcgo{ 0, 0, 0, 0 };
    // grid
    int nx = 6, ny = 6, nz = 3;
    std::vector<int32_t> conn;
    std::vector<double> trans, pv, z;

    Deck()
    {
        const double dx = 10, dy = 10, dz = 2, perm = 1e-13, poro = 0.2;
        auto id = [&](int i, int j, int k) { return (k * ny + j) * nx + i; };
        for (int k = 0; k < nz; ++k) for (int j = 0; j < ny; ++j) for (int i = 0; i < nx; ++i) {
            pv.push_back(poro * dx * dy * dz); z.push_back(2000.0 + (k + 0.5) * dz);
        }
        for (int k = 0; k < nz; ++k) for (int j = 0; j < ny; ++j) for (int i = 0; i + 1 < nx; ++i) { conn.push_back(id(i, j, k)); conn.push_back(id(i + 1, j, k)); trans.push_back(perm * dy * dz / dx); }
        for (int k = 0; k < nz; ++k) for (int j = 0; j + 1 < ny; ++j) for (int i = 0; i < nx; ++i) { conn.push_back(id(i, j, k)); conn.push_back(id(i, j + 1, k)); trans.push_back(perm * dx * dz / dy); }
        for (int k = 0; k + 1 < nz; ++k) for (int j = 0; j < ny; ++j) for (int i = 0; i < nx; ++i) { conn.push_back(id(i, j, k)); conn.push_back(id(i, j, k + 1)); trans.push_back(0.1 * perm * dx * dy / dz); }
    }
    opmgpu_grid grid() const
    {
        opmgpu_grid g{};
        g.nc = nx * ny * nz; g.nconn = int32_t(trans.size()); g.conn_cells = conn.data(); g.trans = trans.data(); g.pv = pv.data(); g.z = z.data();
        g.gravity = 9.80665;
        return g;
    }
    opmgpu_tables tables() const
    {
        opmgpu_tables t{};
        t.n_pvt_regions = 1; t.n_sat_regions = 1; t.has_disgas = 0; t.has_vapoil = 0;
        t.surface_density = dens.data(); t.pvtw = pvtw.data();
        t.oil_node_ptr = oil_node_ptr.data(); t.oil_rs = oil_rs.data(); t.oil_psat = oil_p.data(); t.oil_invb_sat = oil_invb.data(); t.oil_invbmu_sat = oil_invbmu.data();
        t.oil_col_ptr = oil_col_ptr.data(); t.oil_col_p = oil_p.data(); t.oil_col_invb = oil_invb.data(); t.oil_col_invbmu = oil_invbmu.data();
        t.gas_node_ptr = gas_node_ptr.data(); t.gas_pg = gas_p.data(); t.gas_rvsat = gas_rv.data(); t.gas_invb_sat = gas_invb.data(); t.gas_invbmu_sat = gas_invbmu.data();
        t.gas_col_ptr = gas_col_ptr.data(); t.gas_col_rv = gas_rv.data(); t.gas_col_invb = gas_invb.data(); t.gas_col_invbmu = gas_invbmu.data();
        t.swof_ptr = swof_ptr.data(); t.swof_sw = sw.data(); t.swof_krw = krw.data(); t.swof_krow = krow.data(); t.swof_pcow = pcow.data();
        t.sgof_ptr = sgof_ptr.data(); t.sgof_sg = sg.data(); t.sgof_krg = krg.data(); t.sgof_krog = krog.data(); t.sgof_pcgo = pcgo.data();
        t.rock_pref = 200e5; t.rock_comp = 4e-10;
        return t;
    }
};

} // namespace

int main()
{
    std::printf("%s, devices: %d\n", opmgpu_version(), opmgpu_device_count());
    const Deck deck;
    const opmgpu_grid grid = deck.grid();
    const opmgpu_tables tables = deck.tables();
    if (opmgpu_device_count() <= 0) {
        try {
            opmgpu::BlackoilModelGpu model(grid, tables);
            std::printf("host_check: FAILED, a model was created without a device\n");
            return 1;
        } catch (const std::exception& e) {
            std::printf("expected without a GPU: %s\n", e.what());
        }
        return 0;
    }
    const int nc = grid.nc;
    std::vector<double> p(nc), sat(3 * size_t(nc)), rs(nc, 0.0), rv(nc, 0.0);
    std::vector<int8_t> hc(nc, int8_t(OPMGPU_HC_GAS_AND_OIL));
    for (int c = 0; c < nc; ++c) {
        p[c] = 200e5 + 2e5 * std::sin(0.7 * c);                      // out of equilibrium: the step has something to do
        sat[3 * c] = 0.3; sat[3 * c + 2] = (c % 5 == 0) ? 0.1 : 0.0; sat[3 * c + 1] = 1.0 - sat[3 * c] - sat[3 * c + 2];
    }
    opmgpu::ReservoirStateView state{ p.data(), sat.data(), rs.data(), rv.data(), hc.data() };
    try {
        opmgpu::BlackoilModelGpu model(grid, tables);
        opmgpu::NonlinearSolverGpu solver;
        const double p_before = p[0];
        model.prepareStep(86400.0, state);
        const int its = solver.step(model);
        model.downloadState(state);
        double smin = 1.0, ssum_err = 0.0;
        for (int c = 0; c < nc; ++c) {
            for (int a = 0; a < 3; ++a) smin = std::min(smin, sat[3 * c + a]);
            ssum_err = std::max(ssum_err, std::abs(sat[3 * c] + sat[3 * c + 1] + sat[3 * c + 2] - 1.0));
        }
        if (!(smin >= 0.0) || !(ssum_err < 1e-12) || !(p[0] != p_before)) {
            std::printf("host_check: FAILED, implausible state after the step (smin %g, sum error %g)\n", smin, ssum_err);
            return 1;
        }
        std::printf("host_check: converged in %d Newton iterations, last linear solve %d iterations, relaxation %.2f\n", its,
                    model.linearIterationsLastSolve(), model.relaxation());
        // two wells with the device well model: a rate-controlled water injector (full column) and a BHP-controlled producer
        {
            const int32_t connpos[3] = { 0, 3, 6 };
            const int32_t cells[6] = { 0, 36, 72, 35, 71, 107 };
            const double WI[6] = { 2e-12, 2e-12, 2e-12, 2e-12, 2e-12, 2e-12 };
            // controls as WellsManager lays them out: the injector on its rate target with a BHP limit behind it (updateWellControls
            // switches when the limit is broken; it is not, here), the producer on BHP
            const int32_t type[2] = { 0, 1 }, ctrl_ptr[3] = { 0, 2, 3 }, ctrl_type[3] = { OPMGPU_CTRL_SURFACE_RATE, OPMGPU_CTRL_BHP, OPMGPU_CTRL_BHP };
            const double depth_ref[2] = { 2001.0, 2001.0 }, comp_frac[6] = { 1, 0, 0, 0, 1, 0 };
            const double ctrl_target[3] = { 5.0 / 86400.0, 600e5, 150e5 }, ctrl_distr[9] = { 1, 0, 0, 0, 0, 0, 0, 0, 0 };
            opmgpu_wells wells{};
            wells.nw = 2; wells.well_connpos = connpos; wells.well_cells = cells; wells.WI = WI; wells.type = type; wells.depth_ref = depth_ref;
            wells.comp_frac = comp_frac; wells.ctrl_ptr = ctrl_ptr; wells.ctrl_type = ctrl_type; wells.ctrl_target = ctrl_target; wells.ctrl_distr = ctrl_distr;
            model.setDeviceWells(wells);
            const double bhp0[2] = { 1.01 * p[0], 150e5 }, qs0[6] = { 5.0 / 86400.0, 0, 0, 0, 0, 0 };
            model.setWellState(bhp0, qs0);
            const int32_t current0[2] = { 0, 0 };
            model.setWellControls(current0);
        }
        // a 20-day report step through the adaptive sub-stepping loop (AdaptiveTimeStepping::stepImpl), state resident on the device
        opmgpu::AdaptiveTimeSteppingGpu ats;
        ats.step(20 * 86400.0, solver, model);
        double total = 0.0;
        for (double d : ats.substeps) total += d;
        if (ats.substeps.size() < 2 || std::abs(total - 20 * 86400.0) > 1e-6 || !(ats.substeps[1] > ats.substeps[0])) {
            std::printf("host_check: FAILED, adaptive stepping took %zu sub-steps summing to %g s\n", ats.substeps.size(), total);
            return 1;
        }
        double bhp[2], qs[6];
        model.getWellState(bhp, qs);
        if (!(std::abs(qs[0] - 5.0 / 86400.0) < 1e-9) || !(qs[4] < 0.0) || !(std::abs(bhp[1] - 150e5) < 1.0) || !(bhp[0] > 150e5)) {
            std::printf("host_check: FAILED, well state after the report step: inj rate %g bhp %g, prod oil rate %g bhp %g\n", qs[0], bhp[0], qs[4], bhp[1]);
            return 1;
        }
        int32_t current[2] = { -1, -1 };
        model.getWellControls(current);
        if (current[0] != 0 || current[1] != 0) {
            std::printf("host_check: FAILED, well controls after the report step: %d %d (no limit was broken)\n", current[0], current[1]);
            return 1;
        }
        std::printf("host_check: wells: injector bhp %.1f bar at %.2f m3/d water, producer %.2f m3/d oil at %.1f bar\n", bhp[0] / 1e5, qs[0] * 86400.0,
                    -qs[4] * 86400.0, bhp[1] / 1e5);
        // one more 2-day step driven through the one-call Newton iteration (opmgpu_nonlinear_iteration) from the resident state
        {
            opmgpu_newton_ctl ctl{};
            ctl.min_iter = 1; ctl.use_update_stabilization = 1; ctl.relax_type = OPMGPU_RELAX_DAMPEN; ctl.relax_max = 0.5; ctl.relax_increment = 0.1; ctl.relax_rel_tol = 0.2;
            model.prepareStep(2 * 86400.0);
            int it = 0, lin = 0; bool conv = false;
            do { conv = model.nonlinearIterationOneCall(it, true, ctl, &lin); ++it; } while ((!conv && it <= 10) || it <= ctl.min_iter);
            if (!conv) { std::printf("host_check: FAILED, the one-call Newton loop did not converge in %d iterations\n", it); return 1; }
            std::printf("host_check: one-call Newton loop: 2-day step converged in %d iterations\n", it);
        }
        // computeFluidInPlace for the resident state, the field as one region (SimulatorBase_impl.hpp:278)
        {
            const std::vector<int> whole_field;
            const auto fip = model.computeFluidInPlace(whole_field);
            if (fip.size() != 1 || fip[0].size() != 7 || !(fip[0][0] > 0.0) || !(fip[0][1] > 0.0) || !(fip[0][5] > 0.0) || !(fip[0][6] > 1e5)) {
                std::printf("host_check: FAILED, fluids in place: %zu regions\n", fip.size());
                return 1;
            }
            std::printf("host_check: fluids in place: water %.5g, oil %.5g, free gas %.5g sm3, pore volume %.5g m3, hydrocarbon-pv weighted pressure %.1f bar\n",
                        fip[0][0], fip[0][1], fip[0][2], fip[0][5], fip[0][6] / 1e5);
        }
        // RateConverter of SimulatorBase::computeRESV on the resident state: one region, the coefficients of the top PVT region
        {
            opmgpu::SurfaceToReservoirVoidageGpu cvrt(model.handle());
            cvrt.defineState();
            double distr[3] = { 0.0, 0.0, 0.0 };
            cvrt.calcCoeff(0, 0, distr);
            const auto& a = cvrt.attributes(0);
            if (!(a.pressure > 1e5) || !(distr[0] > 0.9 && distr[0] < 1.1) || !(distr[1] > 0.0) || !(distr[2] > 0.0)) {
                std::printf("host_check: FAILED, voidage coefficients %g %g %g at %g Pa\n", distr[0], distr[1], distr[2], a.pressure);
                return 1;
            }
            std::printf("host_check: RESV coefficients at the field's average state (%.1f bar, rs %.3g, rv %.3g): %.5g %.5g %.5g\n",
                        a.pressure / 1e5, a.rs, a.rv, distr[0], distr[1], distr[2]);
        }
        std::printf("host_check: report step of 20 d in %zu sub-steps (first %.2f d, last %.2f d), %d failed, next suggestion %.2f d\n",
                    ats.substeps.size(), ats.substeps.front() / 86400.0, ats.substeps.back() / 86400.0, ats.failed_substeps,
                    ats.suggested_next_timestep / 86400.0);
    } catch (const std::exception& e) {
        std::printf("host_check: FAILED with exception: %s\n", e.what());
        return 1;
    }
    return 0;
}
