// ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle_math.hpp header).  PARITY UNPINNED.
// extern "C" surface of the CPU restatement; mirrors include/vilsolve.h with an orc_ prefix so the
// parity tests drive oracle and HIP library through identical ctypes structures.
#include "oracle.hpp"

namespace orc {
int solve(const vil_problem* p, vil_state* st, const vil_options* o, vil_summary* sum);
int linearize_api(const vil_problem* p, const vil_state* st, const vil_options* o, double* cost, double* S, double* g);
int linearize_full_api(const vil_problem* p, const vil_state* st, const vil_options* o, double* cost, double* Hcc, double* bc, double* hll, double* bl, double* E);
double cost_api(const vil_problem* p, const vil_state* st, const vil_options* o);
int gauge_fix(const double* pose0_before, vil_state* s);
int marginalize(const vil_problem* p, const vil_state* st, const vil_options* o, const vil_marg_spec* spec, vil_prior_out* out);
void sym_eig(int n, const double* Ain, double* w, double* V);
void sym_eig_jacobi(int n, const double* Ain, double* w, double* V);
void set_threads(int n);
#ifdef ORC_TIMERS
extern double g_tm[8];
#endif
}  // namespace orc

extern "C" {

void orc_default_options(vil_options* o) {
    o->max_iterations = 30; o->max_time_s = 0.0;
    o->function_tolerance = 1e-6; o->gradient_tolerance = 1e-10; o->parameter_tolerance = 1e-8;
    o->initial_radius = 1e4; o->max_radius = 1e16; o->min_relative_decrease = 1e-3;
    o->min_mu = 1e-8; o->max_mu = 1.0; o->jacobi_scaling = 1;
    o->visual_loss = VIL_LOSS_CAUCHY; o->visual_loss_scale = 1.0;
    o->lidar_loss = VIL_LOSS_HUBER; o->lidar_loss_scale = 0.1;
    o->rel_loss = VIL_LOSS_CAUCHY; o->rel_loss_scale = 1.0;
    o->autodiff_quirk = 1; o->precision = 0;
}

int orc_eval_factors(const vil_problem* p, const vil_state* s, int cls, double* r, double* J) {
    using namespace orc;
    switch (cls) {
        case VIL_FACTOR_IMU:
            for (int f = 0; f < p->n_imu; ++f) {
                const int i = p->imu_i[f], j = p->imu_j[f];
                imu_evaluate(p->imu_const + (size_t)f * VIL_IMU_CONST, p->G, s->pose + 7 * i, s->speedbias + 9 * i, s->pose + 7 * j, s->speedbias + 9 * j,
                             r + (size_t)f * VIL_IMU_NR, J ? J + (size_t)f * VIL_IMU_NJ : nullptr);
            }
            return VIL_OK;
        case VIL_FACTOR_VISUAL:
            for (int f = 0; f < p->n_vis; ++f)
                visual_evaluate(p->vis_const + (size_t)f * VIL_VIS_CONST, p->sqrt_info_px, p->tr_over_row, p->use_td,
                                s->pose + 7 * p->vis_i[f], s->pose + 7 * p->vis_j[f], s->ex_pose, s->inv_depth[p->vis_l[f]], s->td[0],
                                r + (size_t)f * VIL_VIS_NR, J ? J + (size_t)f * VIL_VIS_NJ : nullptr);
            return VIL_OK;
        case VIL_FACTOR_PRIOR: {
            if (p->prior.n <= 0) return VIL_OK;
            std::vector<const double*> params(p->prior.nblk);
            for (int b = 0; b < p->prior.nblk; ++b) {
                switch (p->prior.blk_kind[b]) {
                    case VIL_BLK_POSE: params[b] = s->pose + 7 * p->prior.blk_index[b]; break;
                    case VIL_BLK_SPEEDBIAS: params[b] = s->speedbias + 9 * p->prior.blk_index[b]; break;
                    case VIL_BLK_EX: params[b] = s->ex_pose; break;
                    default: params[b] = s->td; break;
                }
            }
            prior_evaluate(p->prior, params.data(), r, J);
            return VIL_OK;
        }
        case VIL_FACTOR_ICP:
            for (int f = 0; f < p->n_icp; ++f) {
                const int* id = p->icp_ids + 4 * f;
                icp_evaluate(p->icp_const + (size_t)f * VIL_ICP_CONST, s->pose + 7 * id[0], s->pose + 7 * id[1], s->pose + 7 * id[2], s->pose + 7 * id[3],
                             r + (size_t)f * VIL_ICP_NR, J ? J + (size_t)f * VIL_ICP_NJ : nullptr);
            }
            return VIL_OK;
        case VIL_FACTOR_LPS:
            for (int f = 0; f < p->n_lps; ++f) {
                const int* id = p->lps_ids + 2 * f;
                lps_evaluate(p->lps_const + (size_t)f * VIL_LPS_CONST, s->pose + 7 * id[0], s->pose + 7 * id[1], r + (size_t)f * VIL_LPS_NR, J ? J + (size_t)f * VIL_LPS_NJ : nullptr);
            }
            return VIL_OK;
        case VIL_FACTOR_EDGE:
            for (int f = 0; f < p->n_edge; ++f)
                edge_evaluate(p->edge_const + (size_t)f * VIL_EDGE_CONST, p->q_lb, p->t_lb, s->pose + 7 * p->edge_pose[f], r + (size_t)f * VIL_EDGE_NR, J ? J + (size_t)f * VIL_EDGE_NJ : nullptr);
            return VIL_OK;
        case VIL_FACTOR_PLANE:
            for (int f = 0; f < p->n_plane; ++f)
                plane_evaluate(p->plane_const + (size_t)f * VIL_PLANE_CONST, p->q_lb, p->t_lb, s->pose + 7 * p->plane_pose[f], r + (size_t)f * VIL_PLANE_NR, J ? J + (size_t)f * VIL_PLANE_NJ : nullptr);
            return VIL_OK;
    }
    return VIL_ERR_INVALID_ARGUMENT;
}

// the four functors of lidar_mapping/src/lidarFactor.hpp in window-pose form (include/vilsolve.h: vil_eval_lidar_functors)
int orc_eval_lidar_functors(int32_t kind, int32_t n, const double* c, const double* q_lb, const double* t_lb, const double* pose7, double* r, double* J) {
    using namespace orc;
    for (int f = 0; f < n; ++f) {
        switch (kind) {
            case VIL_LIDAR_EDGE: edge_evaluate(c + 9 * (size_t)f, q_lb, t_lb, pose7, r + 3 * (size_t)f, J ? J + 21 * (size_t)f : nullptr); break;
            case VIL_LIDAR_PLANE_NORM: plane_evaluate(c + 7 * (size_t)f, q_lb, t_lb, pose7, r + (size_t)f, J ? J + 7 * (size_t)f : nullptr); break;
            case VIL_LIDAR_PLANE3: plane3_evaluate(c + 12 * (size_t)f, q_lb, t_lb, pose7, r + (size_t)f, J ? J + 7 * (size_t)f : nullptr); break;
            case VIL_LIDAR_DISTANCE: distance_evaluate(c + 6 * (size_t)f, q_lb, t_lb, pose7, r + 3 * (size_t)f, J ? J + 21 * (size_t)f : nullptr); break;
            default: return VIL_ERR_INVALID_ARGUMENT;
        }
    }
    return VIL_OK;
}

int orc_solve(const vil_problem* p, vil_state* s, const vil_options* o, vil_summary* sum) { return orc::solve(p, s, o, sum); }
int orc_linearize(const vil_problem* p, const vil_state* s, const vil_options* o, double* cost, double* S, double* g) { return orc::linearize_api(p, s, o, cost, S, g); }
int orc_linearize_full(const vil_problem* p, const vil_state* s, const vil_options* o, double* cost, double* Hcc, double* bc, double* hll, double* bl, double* E) { return orc::linearize_full_api(p, s, o, cost, Hcc, bc, hll, bl, E); }
double orc_cost(const vil_problem* p, const vil_state* s, const vil_options* o) { return orc::cost_api(p, s, o); }
int orc_gauge_fix(const double* pose0_before, vil_state* s) { return orc::gauge_fix(pose0_before, s); }
int orc_marginalize(const vil_problem* p, const vil_state* s, const vil_options* o, const vil_marg_spec* spec, vil_prior_out* out) { return orc::marginalize(p, s, o, spec, out); }
void orc_sym_eig(int n, const double* A, double* w, double* V) { orc::sym_eig(n, A, w, V); }
#ifdef ORC_TIMERS
void orc_timers(double* out) { for (int i = 0; i < 8; ++i) { out[i] = orc::g_tm[i]; orc::g_tm[i] = 0; } }
#endif
// all-cores variant of the CPU baseline (bench.py only; parity always runs with 1 thread)
void orc_set_threads(int n) { orc::set_threads(n); }
void orc_sym_eig_jacobi(int n, const double* A, double* w, double* V) { orc::sym_eig_jacobi(n, A, w, V); }
int orc_imu_sqrt_info(const double* cov, double* U) { return orc::imu_sqrt_info(cov, U) ? 0 : -1; }
void orc_loss(int kind, double a, double s, double* rho3) { orc::loss_evaluate(kind, a, s, rho3); }
void orc_edge_residual_ref(const double* cp, const double* a3, const double* b3, const double* q, const double* t, double* r) { orc::edge_residual_ref(cp, a3, b3, q, t, r); }
void orc_plane_residual_ref(const double* cp, const double* n3, double d, const double* q, const double* t, double* r) { orc::plane_residual_ref(cp, n3, d, q, t, r); }

// A5: pre-integrate a sample stream (dt[n], acc[n][3], gyr[n][3]) starting from (acc0, gyr0) -> 287-double record
void orc_preintegrate(int n, const double* dt, const double* acc, const double* gyr, const double* acc0, const double* gyr0,
                      const double* ba, const double* bg, const double* noise4, double* out287) {
    orc::Preint s;
    orc::preint_init(s, acc0, gyr0, ba, bg, noise4);
    for (int i = 0; i < n; ++i) orc::preint_push(s, dt[i], acc + 3 * i, gyr + 3 * i);
    orc::preint_pack(s, out287);
}


// include/vilpreint.h on the CPU: IntegrationBase::repropagate for n intervals; interval k owns samples [start[k], start[k+1])
struct vpre_ctx { int unused; };
int orc_vpre_create(int32_t, vpre_ctx** out) { *out = new vpre_ctx(); return 0; }
void orc_vpre_destroy(vpre_ctx* c) { delete c; }
int orc_vpre_integrate(vpre_ctx*, int32_t n, const int32_t* start, const double* dt, const double* acc, const double* gyr, const double* acc0, const double* gyr0,
                       const double* ba, const double* bg, const double* noise4, double* imu_const, double* jacobian) {
    for (int k = 0; k < n; ++k) {
        orc::Preint s;
        orc::preint_init(s, acc0 + 3 * k, gyr0 + 3 * k, ba + 3 * k, bg + 3 * k, noise4);
        for (int i = start[k]; i < start[k + 1]; ++i) orc::preint_push(s, dt[i], acc + 3 * (size_t)i, gyr + 3 * (size_t)i);
        orc::preint_pack(s, imu_const + (size_t)287 * k);
        if (jacobian) std::memcpy(jacobian + (size_t)225 * k, s.jac, sizeof s.jac);
    }
    return 0;
}

}  // extern "C"
