// se3.h -- SE(3) / projection math shared by the optimizer kernels (fp64, host + device).
// Follows g2o's SE3Quat (reference src/g2o_catkin/include/g2o/types/se3quat.h:58-60,
// 223-257, 280-285) and Eigen's quaternion <-> matrix conversions.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>

#define ASD_HD __host__ __device__ inline

// 1/sqrt(x) for the pose kernels.  On the device: hardware seed + two Newton steps (~1 ulp) -- PoseOptimization runs
// its 6x6 solves and pose updates on ONE lane, where every dependent fp64 instruction costs ~44 cycles and the IEEE
// expansions of sqrt and division are chains of 12-17 of them.  On the host: the plain expression.
ASD_HD double asd_rsqrt(double x) {
#ifdef __HIP_DEVICE_COMPILE__
  double r = __builtin_amdgcn_rsq(x);
  r = fma(fma(-0.5 * x * r, r, 0.5), r, r);
  r = fma(fma(-0.5 * x * r, r, 0.5), r, r);
  return r;
#else
  return 1.0 / sqrt(x);
#endif
}

struct Pose7 {  // unit quaternion (x,y,z,w), translation: world -> camera
  double qx, qy, qz, qw, tx, ty, tz;
};

ASD_HD void quat_normalize(double& x, double& y, double& z, double& w) {  // normalizeRotation
  if (w < 0) { x = -x; y = -y; z = -z; w = -w; }
  const double inv = asd_rsqrt(x * x + y * y + z * z + w * w);  // Eigen's `/= norm()` multiplies by the reciprocal
  x *= inv; y *= inv; z *= inv; w *= inv;
}

ASD_HD void quat_to_rot(const Pose7& T, double R[9]) {
  const double tx = 2 * T.qx, ty = 2 * T.qy, tz = 2 * T.qz;
  const double twx = tx * T.qw, twy = ty * T.qw, twz = tz * T.qw;
  const double txx = tx * T.qx, txy = ty * T.qx, txz = tz * T.qx, tyy = ty * T.qy, tyz = tz * T.qy, tzz = tz * T.qz;
  R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
  R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}

ASD_HD void quat_rotate(double qx, double qy, double qz, double qw, const double v[3], double out[3]) {
  double ux = qy * v[2] - qz * v[1], uy = qz * v[0] - qx * v[2], uz = qx * v[1] - qy * v[0];
  ux += ux; uy += uy; uz += uz;
  out[0] = v[0] + qw * ux + (qy * uz - qz * uy);
  out[1] = v[1] + qw * uy + (qz * ux - qx * uz);
  out[2] = v[2] + qw * uz + (qx * uy - qy * ux);
}

ASD_HD void pose_map(const Pose7& T, const double X[3], double out[3]) {  // SE3Quat::map
  quat_rotate(T.qx, T.qy, T.qz, T.qw, X, out);
  out[0] += T.tx; out[1] += T.ty; out[2] += T.tz;
}

// VertexSE3Expmap::oplusImpl: T <- SE3Quat::exp(u) * T, u = (omega, upsilon)   (se3quat.h:223-257, types_six_dof_expmap.h:94-97)
// This runs on ONE lane between two passes of PoseOptimization, so it is written for a short chain of dependent fp64 instructions:
//  * theta < 1e-5 (the reference's small-angle branch, kept as is: R = I + W + W^2, V = R, then Quaterniond(R)) -- no 1/theta needed;
//  * theta < 0.1: everything as series in theta^2 (seven terms: truncation < 1e-19 relative): the rotation's quaternion directly as
//    (omega sin(theta/2)/theta, cos(theta/2)) -- the same rotation the reference reaches through R = I + a W + b W^2 and
//    Quaterniond(R), without the trace / square root / normalisation chain and without the cancellation of 1 - cos theta -- and
//    V = I + b W + c W^2 with b = (1 - cos t)/t^2, c = (t - sin t)/t^3;
//  * beyond: the closed forms through sincos, as the reference.
ASD_HD void rot_to_quat_eigen(const double R[9], double& ex, double& ey, double& ez, double& ew) {   // Eigen::Quaterniond(Matrix3d)
  const double tr = R[0] + R[4] + R[8];
  if (tr > 0) {
    const double rs = asd_rsqrt(tr + 1.0);
    ew = 0.5 * ((tr + 1.0) * rs);
    const double s = 0.5 * rs;
    ex = (R[7] - R[5]) * s; ey = (R[2] - R[6]) * s; ez = (R[3] - R[1]) * s;
  } else {
    // largest diagonal element first (Eigen quaternionbase_assign_impl); written out per case so every
    // index is static (a runtime-indexed R[] would live in scratch memory on the GPU)
    int i = 0;
    if (R[4] > R[0]) i = 1;
    if (R[8] > (i == 0 ? R[0] : R[4])) i = 2;
    if (i == 0) {
      double s = sqrt(R[0] - R[4] - R[8] + 1.0);
      ex = 0.5 * s; s = 0.5 / s;
      ew = (R[7] - R[5]) * s; ey = (R[3] + R[1]) * s; ez = (R[6] + R[2]) * s;
    } else if (i == 1) {
      double s = sqrt(R[4] - R[8] - R[0] + 1.0);
      ey = 0.5 * s; s = 0.5 / s;
      ew = (R[2] - R[6]) * s; ez = (R[7] + R[5]) * s; ex = (R[1] + R[3]) * s;
    } else {
      double s = sqrt(R[8] - R[0] - R[4] + 1.0);
      ez = 0.5 * s; s = 0.5 / s;
      ew = (R[3] - R[1]) * s; ex = (R[2] + R[6]) * s; ey = (R[5] + R[7]) * s;
    }
  }
}

ASD_HD Pose7 pose_oplus(const Pose7& T, const double u[6]) {
  const double wx = u[0], wy = u[1], wz = u[2];
  const double theta2 = wx * wx + wy * wy + wz * wz;
  const double O[9] = {0, -wz, wy, wz, 0, -wx, -wy, wx, 0};
  double O2[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) O2[i * 3 + j] = O[i * 3] * O[j] + O[i * 3 + 1] * O[3 + j] + O[i * 3 + 2] * O[6 + j];
  double V[9];
  double ex, ey, ez, ew;
  if (theta2 < 1e-10) {  // theta < 0.00001, se3quat.h:237-243 (kept as is: R = I + W + W^2, V = R)
    double R[9];
    for (int i = 0; i < 9; ++i) { R[i] = (i % 4 == 0 ? 1.0 : 0.0) + O[i] + O2[i]; V[i] = R[i]; }
    rot_to_quat_eigen(R, ex, ey, ez, ew);
    quat_normalize(ex, ey, ez, ew);
  } else if (theta2 < 0.01) {
    const double q = theta2;
    const double b = 0.5 + q * (-1.0 / 24 + q * (1.0 / 720 + q * (-1.0 / 40320 + q * (1.0 / 3628800 + q * (-1.0 / 479001600 + q * (1.0 / 87178291200.0))))));
    const double c = 1.0 / 6 + q * (-1.0 / 120 + q * (1.0 / 5040 + q * (-1.0 / 362880 + q * (1.0 / 39916800 + q * (-1.0 / 6227020800.0 + q * (1.0 / 1307674368000.0))))));
    const double sh = 0.5 + q * (-1.0 / 48 + q * (1.0 / 3840 + q * (-1.0 / 645120 + q * (1.0 / 185794560 + q * (-1.0 / 81749606400.0 + q * (1.0 / 51011754393600.0))))));
    const double ch = 1.0 + q * (-1.0 / 8 + q * (1.0 / 384 + q * (-1.0 / 46080 + q * (1.0 / 10321920 + q * (-1.0 / 3715891200.0 + q * (1.0 / 1961990553600.0))))));
    for (int i = 0; i < 9; ++i) V[i] = (i % 4 == 0 ? 1.0 : 0.0) + b * O[i] + c * O2[i];
    ex = wx * sh; ey = wy * sh; ez = wz * sh; ew = ch;   // unit up to rounding (ch > 0: the sign convention of normalizeRotation holds)
  } else {
    const double itheta = asd_rsqrt(theta2);
    const double theta = theta2 * itheta;
    double sn, cs;
    sincos(theta, &sn, &cs);
    const double it2 = itheta * itheta;
    const double a = sn * itheta, b = (1 - cs) * it2;
    const double c = (theta - sn) * it2 * itheta;
    double R[9];
    for (int i = 0; i < 9; ++i) {
      const double I = (i % 4 == 0 ? 1.0 : 0.0);
      R[i] = I + a * O[i] + b * O2[i];
      V[i] = I + b * O[i] + c * O2[i];
    }
    rot_to_quat_eigen(R, ex, ey, ez, ew);
    quat_normalize(ex, ey, ez, ew);
  }
  const double et[3] = {V[0] * u[3] + V[1] * u[4] + V[2] * u[5], V[3] * u[3] + V[4] * u[4] + V[5] * u[5],
                        V[6] * u[3] + V[7] * u[4] + V[8] * u[5]};
  const double tt[3] = {T.tx, T.ty, T.tz};
  double rt[3];
  quat_rotate(ex, ey, ez, ew, tt, rt);
  Pose7 o;
  o.tx = et[0] + rt[0]; o.ty = et[1] + rt[1]; o.tz = et[2] + rt[2];
  o.qx = ew * T.qx + ex * T.qw + ey * T.qz - ez * T.qy;
  o.qy = ew * T.qy + ey * T.qw + ez * T.qx - ex * T.qz;
  o.qz = ew * T.qz + ez * T.qw + ex * T.qy - ey * T.qx;
  o.qw = ew * T.qw - ex * T.qx - ey * T.qy - ez * T.qz;
  quat_normalize(o.qx, o.qy, o.qz, o.qw);
  return o;
}

// RobustKernelHuber::robustify (robust_kernel_impl.cpp:78-91): rho0 = cost, rho1 = weight
ASD_HD void huber(double e2, double delta, double& rho0, double& rho1) {
  const double dsqr = delta * delta;
  if (e2 <= dsqr) { rho0 = e2; rho1 = 1.0; }
  else { const double s = sqrt(e2); rho0 = 2 * s * delta - dsqr; rho1 = delta / s; }
}

// EdgeSE3ProjectXYZ(OnlyPose) pose Jacobian (types_six_dof_expmap.cpp:136-150): rows [omega | upsilon]
ASD_HD void jac_pose(double x, double y, double z, double fx, double fy, double J[12]) {
  const double iz = 1.0 / z, iz2 = iz * iz;
  J[0] = x * y * iz2 * fx; J[1] = -(1 + (x * x * iz2)) * fx; J[2] = y * iz * fx;
  J[3] = -iz * fx; J[4] = 0; J[5] = x * iz2 * fx;
  J[6] = (1 + y * y * iz2) * fy; J[7] = -x * y * iz2 * fy; J[8] = -x * iz * fy;
  J[9] = 0; J[10] = -iz * fy; J[11] = y * iz2 * fy;
}
