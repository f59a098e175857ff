// gf_preint.hpp -- running state of an IMU pre-integration (IntegrationBase, factor/integration_base.h:22-62) for the host code that appends samples
// to an interval frame after frame (gf_estimator.hip): integrating only the new samples repeats exactly the operations of starting over.
#pragma once
#include <vector>
namespace gf {
struct ImuPreState {
    double acc_0[3], gyr_0[3], dp[3], dq[4], dv[3], J[225], P[225], sum_dt;
    int n_done;
};
void imu_preint_reset(ImuPreState& st, const double* acc0, const double* gyr0);
void imu_preint_range(ImuPreState& st, const double* ba, const double* bg, const double* noise, const double* dt, const double* acc, const double* gyr, int s0, int s1);
// delta_p / delta_q / delta_v / sum_dt only (J and P of `st` are left alone): the same bits as imu_preint_range gives for them
void imu_preint_state_range(ImuPreState& st, const double* ba, const double* bg, const double* dt, const double* acc, const double* gyr, int s0, int s1);
// many intervals at once on the device (gf_preint.hip): every job is integrated from scratch over its n samples and fills *st with the bits the two calls
// above would produce
struct PreintBatch;
struct PreintJob { ImuPreState* st; const double *ba, *bg, *dt, *acc, *gyr; int n; double acc0[3], gyr0[3]; };
int preint_batch_create(PreintBatch** out);
void preint_batch_destroy(PreintBatch* b);
int preint_batch_run(PreintBatch* b, const std::vector<PreintJob>& jobs, const double* noise);
}  // namespace gf
