// camera.hip — Camera::CamFromImg of all keypoints of an image, once per image (camera_math.h has
// the why): one thread per keypoint, the Newton iteration of IterativeUndistortion in FP64 without
// contraction, so the lifted coordinates equal the host's (and the oracle's) bit for bit.  Only the
// polynomial distortion models come here; the models that need libm are lifted on the host.
// Work per keypoint: <= 100 iterations x 5 distortion evaluations (3-5 iterations in practice);
// bytes: 8 or 16 B read, 16 B written per keypoint - a few microseconds per image, off the hot path.
#include <hip/hip_runtime.h>

#include "amc_internal.h"
#include "camera_math.h"

namespace amc {

__global__ __launch_bounds__(256) void undistort_kernel(const float* __restrict__ kp, const double* __restrict__ kp64,
                                                        uint32_t rows, CameraDev cam, double* __restrict__ kpn) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows) return;
    const double x = kp64 ? kp64[2 * (size_t)i] : (double)kp[2 * (size_t)i];
    const double y = kp64 ? kp64[2 * (size_t)i + 1] : (double)kp[2 * (size_t)i + 1];
    double u, v;
    cam::cam_from_img(cam.model_id, cam.params, x, y, u, v);
    kpn[2 * (size_t)i] = u;
    kpn[2 * (size_t)i + 1] = v;
}

hipError_t launch_undistort(const float* kp, const double* kp64, uint32_t rows, const CameraDev& cam, double* kpn,
                            hipStream_t s) {
    if (rows == 0) return hipSuccess;
    hipLaunchKernelGGL(undistort_kernel, dim3((rows + 255) / 256), dim3(256), 0, s, kp, kp64, rows, cam, kpn);
    return hipGetLastError();
}

// Camera::ImgFromCam of n points of the normalised image plane (polynomial models; the libm models run on the host)
__global__ __launch_bounds__(256) void project_kernel(const double* __restrict__ uv, uint32_t n, CameraDev cam, double* __restrict__ xy) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double x, y;
    cam::img_from_cam(cam.model_id, cam.params, uv[2 * (size_t)i], uv[2 * (size_t)i + 1], x, y);
    xy[2 * (size_t)i] = x;
    xy[2 * (size_t)i + 1] = y;
}
hipError_t launch_project(const double* uv, uint32_t n, const CameraDev& cam, double* xy, hipStream_t s) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(project_kernel, dim3((n + 255) / 256), dim3(256), 0, s, uv, n, cam, xy);
    return hipGetLastError();
}

}  // namespace amc
