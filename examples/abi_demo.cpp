// abi_demo.cpp -- a client of libgs_amd.so that knows nothing about PyTorch or Python: the drop-in boundary
// is the C ABI of include/gs_abi.h (plain pointers and sizes), this is what a non-Python host would write.
//
//   abi_demo <scene.bin> <out_image.bin>
// scene.bin : int32 N, W, H; float fx, fy, near; float rot[9], tran[3]; then pos[N,3] quat[N,4] scale[N,3]
//             opa[N] rgb[N,3] (fp32, raw parameters as the reference stores them)
// out       : int64 visible, pairs; float image[H,W,3]
// Build: hipcc --offload-arch=gfx950 -O2 -I include examples/abi_demo.cpp -L 3d-gaussian-splatting_amd/csrc
//        -lgs_amd -Wl,-rpath,'$ORIGIN/../3d-gaussian-splatting_amd/csrc' -o examples/abi_demo
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <cstring>
#include <vector>

#include "gs_abi.h"

#define HIP_OK(x)                                                            \
    do {                                                                     \
        hipError_t e_ = (x);                                                 \
        if (e_ != hipSuccess) {                                              \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));          \
            return 2;                                                        \
        }                                                                    \
    } while (0)

template <typename T>
static bool read_n(FILE *f, T *dst, size_t n) { return fread(dst, sizeof(T), n, f) == n; }

int main(int argc, char **argv) {
    if (argc != 3) {
        fprintf(stderr, "usage: %s scene.bin out.bin\n", argv[0]);
        return 1;
    }
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 1;
    int32_t hdr[3];
    float cam[3 + 9 + 3];
    if (!read_n(f, hdr, 3) || !read_n(f, cam, 15)) return 1;
    const int64_t N = hdr[0];
    const int W = hdr[1], H = hdr[2];
    std::vector<float> pos(N * 3), quat(N * 4), scale(N * 3), opa(N), rgb(N * 3);
    if (!read_n(f, pos.data(), pos.size()) || !read_n(f, quat.data(), quat.size()) ||
        !read_n(f, scale.data(), scale.size()) || !read_n(f, opa.data(), opa.size()) ||
        !read_n(f, rgb.data(), rgb.size()))
        return 1;
    fclose(f);

    float *d_pos, *d_quat, *d_scale, *d_opa, *d_rgb, *d_img;
    HIP_OK(hipMalloc(&d_pos, pos.size() * 4));
    HIP_OK(hipMalloc(&d_quat, quat.size() * 4));
    HIP_OK(hipMalloc(&d_scale, scale.size() * 4));
    HIP_OK(hipMalloc(&d_opa, opa.size() * 4));
    HIP_OK(hipMalloc(&d_rgb, rgb.size() * 4));
    HIP_OK(hipMalloc(&d_img, (size_t)W * H * 3 * 4));
    HIP_OK(hipMemcpy(d_pos, pos.data(), pos.size() * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_quat, quat.data(), quat.size() * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_scale, scale.data(), scale.size() * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_opa, opa.data(), opa.size() * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_rgb, rgb.data(), rgb.size() * 4, hipMemcpyHostToDevice));

    gs_frame fr = {};
    fr.N = N;
    fr.color_dim = 3;
    fr.scale_activation = 0;
    fr.pos = d_pos;
    fr.quat = d_quat;
    fr.scale = d_scale;
    fr.opa = d_opa;
    fr.rgb = d_rgb;
    const double fx = cam[0], fy = cam[1];
    for (int i = 0; i < 9; ++i) fr.rot[i] = cam[3 + i];
    for (int i = 0; i < 3; ++i) fr.tran[i] = cam[12 + i];
    fr.near_plane = cam[2];
    fr.width = W;
    fr.height = H;
    fr.focal_x = (float)fx;
    fr.focal_y = (float)fy;
    // the scalars of splatter.Tiles (splatter.py:259-282) and the culling guard band (:532-533), in double
    const int padW = (W + 15) / 16 * 16, padH = (H + 15) / 16 * 16;
    fr.tile_length_x = (float)(16.0 / fx);
    fr.tile_length_y = (float)(16.0 / fy);
    fr.leftmost = (float)(-padW / 2.0 / fx);
    fr.topmost = (float)(-padH / 2.0 / fy);
    fr.half_width = (float)(W * 1.2 / 2 / fx);
    fr.half_height = (float)(H * 1.2 / 2 / fy);
    fr.thresh = 0.05f;
    fr.max_pairs = 8 * N + 4096;
    fr.sort_mode = 2;
    fr.tile_culling_method = 2;  // "prob2", train.py's default (0 = "dist", 1 = "prob": splatter.py:571)
    fr.training = 0;
    fr.workspace_bytes = gs_frame_workspace_bytes(N, fr.max_pairs, W, H, 3, 0);
    HIP_OK(hipMalloc(&fr.workspace, fr.workspace_bytes));
    fr.image = d_img;
    hipStream_t s;
    HIP_OK(hipStreamCreate(&s));
    int rc = gs_frame_forward(&fr, s);
    if (rc) {
        fprintf(stderr, "gs_frame_forward: %d (%s)\n", rc, gs_last_error());
        return 3;
    }
    int64_t stats[4];
    rc = gs_frame_stats_async(&fr, stats, s);
    HIP_OK(hipStreamSynchronize(s));
    if (rc || stats[2]) {
        fprintf(stderr, "stats rc %d overflow %lld\n", rc, (long long)stats[2]);
        return 3;
    }
    std::vector<float> img((size_t)W * H * 3);
    HIP_OK(hipMemcpy(img.data(), d_img, img.size() * 4, hipMemcpyDeviceToHost));
    // The same pose once more with GS_FRAME_OCCLUSION_CULL (include/gs_abi.h): the forward above left, per tile, the depth
    // behind which nothing was composited; this frame drops the pairs (and skips the projection of the Gaussians) behind it.
    // The image must be the first one's bit for bit -- a frame whose trimmed lists prove too short is rendered again from
    // the full ones inside the call.  (The library ignores the flag where the cull does not apply: small scenes.)
    {
        gs_frame fc = fr;
        fc.flags |= GS_FRAME_OCCLUSION_CULL;
        int32_t culled = 0;
        rc = gs_frame_is_occlusion_culled(&fc, &culled);
        if (!rc) rc = gs_frame_forward(&fc, s);
        int64_t st2[4] = {0, 0, 0, 0}, fell = 0;
        if (!rc) rc = gs_frame_stats_async(&fc, st2, s);
        if (!rc) rc = gs_frame_cull_fallback_async(&fc, &fell, s);
        HIP_OK(hipStreamSynchronize(s));
        std::vector<float> img2(img.size());
        HIP_OK(hipMemcpy(img2.data(), d_img, img2.size() * 4, hipMemcpyDeviceToHost));
        if (rc || memcmp(img.data(), img2.data(), img.size() * 4) != 0) {
            fprintf(stderr, "occlusion-culled frame: rc %d (%s), image %s\n", rc, gs_last_error(),
                    rc ? "-" : "differs from the unculled frame");
            return 4;
        }
        printf("abi_demo: second frame with GS_FRAME_OCCLUSION_CULL: culled by the library %d, pairs emitted %lld of %lld, "
               "fell back %lld, image identical\n", (int)culled, (long long)st2[1], (long long)stats[1], (long long)fell);
    }
    FILE *o = fopen(argv[2], "wb");
    if (!o) return 1;
    fwrite(stats, sizeof(int64_t), 2, o);
    fwrite(img.data(), 4, img.size(), o);
    fclose(o);
    printf("abi_demo: ABI v%d, N=%lld visible=%lld pairs=%lld\n", gs_abi_version(), (long long)N, (long long)stats[0],
           (long long)stats[1]);
    return 0;
}
