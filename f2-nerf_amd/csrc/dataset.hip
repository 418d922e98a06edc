// Ray generation on gfx950: pixel -> world ray with the reference's Newton undistortion
// (Dataset/Dataset.cu:13-123), plus the pixel gather that turns resident images into a ground-truth colour batch.
// One ray per lane: ~150 flops and at most a few Newton iterations per ray; the point of having it on the device is
// residency -- the reference draws camera / pixel indices on the CPU and uploads rays and colours every iteration
// (Dataset.cpp:275-298), which sits directly in front of a 2 ms training step.
#include "f2n_dev.h"

__device__ __forceinline__ void f2n_distort(const float* k, float u, float v, float& du, float& dv) {  // :13-27
  const float k1 = k[0], k2 = k[1], p1 = k[2], p2 = k[3];
  const float u2 = u * u, uv = u * v, v2 = v * v;
  const float r2 = u2 + v2;
  const float radial = k1 * r2 + k2 * r2 * r2;
  du = u * radial + 2.f * p1 * uv + p2 * (r2 + 2.f * u2);
  dv = v * radial + 2.f * p2 * uv + p1 * (r2 + 2.f * v2);
}

// iterative_camera_undistortion, Dataset.cu:30-72: Newton with central differences; Eigen's 2x2 inverse spelled out
// (invdet = 1/det, det = m00*m11 - m10*m01) so that the oracle and this kernel agree bit for bit.
__device__ __forceinline__ void f2n_undistort(const float* k, float& u, float& v) {
  const float eps = 1.1920928955078125e-07f;
  const float x0[2] = {u, v};
  float x[2] = {u, v};
  for (int it = 0; it < 100; it++) {
    const float step0 = fmaxf(eps, fabsf(1e-6f * x[0]));
    const float step1 = fmaxf(eps, fabsf(1e-6f * x[1]));
    float dx[2], d0b[2], d0f[2], d1b[2], d1f[2];
    f2n_distort(k, x[0], x[1], dx[0], dx[1]);
    f2n_distort(k, x[0] - step0, x[1], d0b[0], d0b[1]);
    f2n_distort(k, x[0] + step0, x[1], d0f[0], d0f[1]);
    f2n_distort(k, x[0], x[1] - step1, d1b[0], d1b[1]);
    f2n_distort(k, x[0], x[1] + step1, d1f[0], d1f[1]);
    const float j00 = 1.f + (d0f[0] - d0b[0]) / (2.f * step0);
    const float j01 = (d1f[0] - d1b[0]) / (2.f * step1);
    const float j10 = (d0f[1] - d0b[1]) / (2.f * step0);
    const float j11 = 1.f + (d1f[1] - d1b[1]) / (2.f * step1);
    const float invdet = 1.f / (j00 * j11 - j10 * j01);
    const float i00 = j11 * invdet, i10 = -j10 * invdet, i01 = -j01 * invdet, i11 = j00 * invdet;
    const float r0 = x[0] + dx[0] - x0[0], r1 = x[1] + dx[1] - x0[1];
    const float s0 = i00 * r0 + i01 * r1, s1 = i10 * r0 + i11 * r1;
    x[0] -= s0;
    x[1] -= s1;
    if (s0 * s0 + s1 * s1 < 1e-10f) break;
  }
  u = x[0];
  v = x[1];
}

// Img2WorldRayKernel, Dataset.cu:93-123, with the half-pixel shift of :126 applied here.
__global__ void img2world_kernel(int n_rays, const float* __restrict__ poses, const float* __restrict__ intri,
                                 const float* __restrict__ dist, const int32_t* __restrict__ cam_idx,
                                 const int32_t* __restrict__ ij, float* __restrict__ rays_o, float* __restrict__ rays_d) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_rays) return;
  const int c = cam_idx[r];
  const float* K = intri + 9 * (size_t) c;
  const float* P = poses + 12 * (size_t) c;
  const float i = (float) ij[2 * r] + .5f, j = (float) ij[2 * r + 1] + .5f;
  const float cx = K[2], cy = K[5], fx = K[0], fy = K[4];
  float u = (j - cx) / fx;
  float v = (i - cy) / fy;  // OpenCV style
  float k[4] = {dist[4 * (size_t) c], dist[4 * (size_t) c + 1], dist[4 * (size_t) c + 2], dist[4 * (size_t) c + 3]};
  f2n_undistort(k, u, v);
  const float dir[3] = {u, -v, -1.f};  // OpenGL style
#pragma unroll
  for (int a = 0; a < 3; a++) {
    rays_d[3 * (size_t) r + a] = f2n_sum3(P[4 * a] * dir[0], P[4 * a + 1] * dir[1], P[4 * a + 2] * dir[2]);
    rays_o[3 * (size_t) r + a] = P[4 * a + 3];
  }
}

// gt_colors[r] = images[cam][i][j] and bounds[r] = cam_bounds[cam] (Dataset.cpp:291-295): resident images, no upload.
__global__ void gather_pixels_kernel(int n_rays, int height, int width, const float* __restrict__ images,
                                     const float* __restrict__ cam_bounds, const int32_t* __restrict__ cam_idx,
                                     const int32_t* __restrict__ ij, float* __restrict__ colors, float* __restrict__ bounds) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_rays) return;
  const int c = cam_idx[r];
  if (colors != nullptr) {
    const size_t px = ((size_t) c * height + ij[2 * r]) * width + ij[2 * r + 1];
#pragma unroll
    for (int a = 0; a < 3; a++) colors[3 * (size_t) r + a] = images[3 * px + a];
  }
  if (bounds != nullptr) {
    bounds[2 * (size_t) r] = cam_bounds[2 * (size_t) c];
    bounds[2 * (size_t) r + 1] = cam_bounds[2 * (size_t) c + 1];
  }
}

// Dataset::RandRaysData (Dataset.cpp:275-298) in ONE launch: three uniforms per ray pick an image of the set and a pixel, the
// ray is generated (img2world_kernel's arithmetic) and the ground-truth colour / bounds gathered.  The ATen spelling -- three
// randint, an index, a stack, then the two kernels above -- was eight dependent launches on the training step's main queue,
// once per iteration.
__global__ void draw_ray_batch_kernel(int n_rays, const float* __restrict__ u01, unsigned long long key, unsigned long long seq,
                                      const int32_t* __restrict__ image_set, int n_set,
                                      int height, int width, const float* __restrict__ poses, const float* __restrict__ intri,
                                      const float* __restrict__ dist, const float* __restrict__ images,
                                      const float* __restrict__ cam_bounds, int32_t* __restrict__ cam_idx, int32_t* __restrict__ ij,
                                      float* __restrict__ rays_o, float* __restrict__ rays_d, float* __restrict__ colors,
                                      float* __restrict__ bounds) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_rays) return;
  float u3[3];
  if (u01 != nullptr) {
    u3[0] = u01[3 * (size_t) r]; u3[1] = u01[3 * (size_t) r + 1]; u3[2] = u01[3 * (size_t) r + 2];
  } else {  // keyed: ray r of batch `seq` draws its own three uniforms (no rand launch in front of this kernel)
    uint32_t x[4];
    f2n_philox4x32((uint32_t) r, 0u, (uint32_t) seq, (uint32_t) (seq >> 32), (uint32_t) key, (uint32_t) (key >> 32), x);
    u3[0] = f2n_u01(x[0]); u3[1] = f2n_u01(x[1]); u3[2] = f2n_u01(x[2]);
  }
  const int c = image_set[min((int) (u3[0] * (float) n_set), n_set - 1)];
  const int pi = min((int) (u3[1] * (float) height), height - 1);
  const int pj = min((int) (u3[2] * (float) width), width - 1);
  cam_idx[r] = c;
  ij[2 * r] = pi;
  ij[2 * r + 1] = pj;
  const float* K = intri + 9 * (size_t) c;
  const float* P = poses + 12 * (size_t) c;
  const float i = (float) pi + .5f, j = (float) pj + .5f;
  const float cx = K[2], cy = K[5], fx = K[0], fy = K[4];
  float u = (j - cx) / fx;
  float v = (i - cy) / fy;
  float k[4] = {dist[4 * (size_t) c], dist[4 * (size_t) c + 1], dist[4 * (size_t) c + 2], dist[4 * (size_t) c + 3]};
  f2n_undistort(k, u, v);
  const float dir[3] = {u, -v, -1.f};
#pragma unroll
  for (int a = 0; a < 3; a++) {
    rays_d[3 * (size_t) r + a] = f2n_sum3(P[4 * a] * dir[0], P[4 * a + 1] * dir[1], P[4 * a + 2] * dir[2]);
    rays_o[3 * (size_t) r + a] = P[4 * a + 3];
  }
  if (colors != nullptr) {
    const size_t px = ((size_t) c * height + pi) * width + pj;
#pragma unroll
    for (int a = 0; a < 3; a++) colors[3 * (size_t) r + a] = images[3 * px + a];
  }
  bounds[2 * (size_t) r] = cam_bounds[2 * (size_t) c];
  bounds[2 * (size_t) r + 1] = cam_bounds[2 * (size_t) c + 1];
}

extern "C" {

int f2n_draw_ray_batch(void* stream, int n_rays, const float* u01, const int32_t* image_set, int n_set, int height, int width,
                       const float* poses, const float* intri, const float* dist_params, const float* images, const float* cam_bounds,
                       int32_t* cam_indices, int32_t* ij, float* rays_o, float* rays_d, float* gt_colors, float* bounds) {
  if (n_rays < 0 || n_set < 1 || height <= 0 || width <= 0 || (gt_colors != nullptr && images == nullptr) || cam_bounds == nullptr || u01 == nullptr)
    return F2N_ERR_INVALID_ARG;
  if (n_rays == 0) return F2N_OK;
  hipLaunchKernelGGL(draw_ray_batch_kernel, dim3(f2n_div_up(n_rays, 256)), dim3(256), 0, (hipStream_t) stream, n_rays, u01, 0ull, 0ull, image_set,
                     n_set, height, width, poses, intri, dist_params, images, cam_bounds, cam_indices, ij, rays_o, rays_d, gt_colors,
                     bounds);
  return f2n_launch_status();
}

int f2n_draw_ray_batch_keyed(void* stream, int n_rays, uint64_t key, uint64_t seq, const int32_t* image_set, int n_set, int height, int width,
                             const float* poses, const float* intri, const float* dist_params, const float* images, const float* cam_bounds,
                             int32_t* cam_indices, int32_t* ij, float* rays_o, float* rays_d, float* gt_colors, float* bounds) {
  if (n_rays < 0 || n_set < 1 || height <= 0 || width <= 0 || (gt_colors != nullptr && images == nullptr) || cam_bounds == nullptr)
    return F2N_ERR_INVALID_ARG;
  if (n_rays == 0) return F2N_OK;
  hipLaunchKernelGGL(draw_ray_batch_kernel, dim3(f2n_div_up(n_rays, 256)), dim3(256), 0, (hipStream_t) stream, n_rays, (const float*) nullptr,
                     (unsigned long long) key, (unsigned long long) seq, image_set, n_set, height, width, poses, intri, dist_params, images,
                     cam_bounds, cam_indices, ij, rays_o, rays_d, gt_colors, bounds);
  return f2n_launch_status();
}

int f2n_img2world_rays(void* stream, int n_rays, const float* poses, const float* intri, const float* dist_params,
                       const int32_t* cam_indices, const int32_t* ij, float* rays_o, float* rays_d) {
  if (n_rays < 0) return F2N_ERR_INVALID_ARG;
  if (n_rays == 0) return F2N_OK;
  hipLaunchKernelGGL(img2world_kernel, dim3(f2n_div_up(n_rays, 256)), dim3(256), 0, (hipStream_t) stream, n_rays, poses, intri,
                     dist_params, cam_indices, ij, rays_o, rays_d);
  return f2n_launch_status();
}

int f2n_gather_pixels(void* stream, int n_rays, int height, int width, const float* images, const float* cam_bounds,
                      const int32_t* cam_indices, const int32_t* ij, float* gt_colors, float* bounds) {
  if (n_rays < 0 || height <= 0 || width <= 0) return F2N_ERR_INVALID_ARG;
  if (gt_colors != nullptr && images == nullptr) return F2N_ERR_INVALID_ARG;
  if (bounds != nullptr && cam_bounds == nullptr) return F2N_ERR_INVALID_ARG;
  if (n_rays == 0) return F2N_OK;
  hipLaunchKernelGGL(gather_pixels_kernel, dim3(f2n_div_up(n_rays, 256)), dim3(256), 0, (hipStream_t) stream, n_rays, height, width,
                     images, cam_bounds, cam_indices, ij, gt_colors, bounds);
  return f2n_launch_status();
}

}  // extern "C"
