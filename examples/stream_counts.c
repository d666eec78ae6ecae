/* examples/stream_counts.c -- the C ABI from plain C: a sequence of gray frames in host memory through Harris, FAST-9
 * and Canny with imgfd_stream_* (include/imgfd.h), printing the per-frame feature counts.
 *
 *   gcc -std=c99 -O2 -Iinclude examples/stream_counts.c -o stream_counts image_amd/libimgfd.so -Wl,-rpath,$PWD/image_amd
 *   ./stream_counts [nx ny n_frames]
 *
 * This is the loop an R user writes as  for (f in files) { image_harris(x); image_detect_corners(x); ... }  moved below
 * the boundary so that the upload of one batch overlaps the kernels of the previous one. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "imgfd.h"

#define CHECK(call)                                                                         \
    do {                                                                                    \
        imgfd_status st_ = (call);                                                          \
        if (st_ != IMGFD_OK) {                                                              \
            fprintf(stderr, "%s -> %d: %s\n", #call, (int)st_, imgfd_last_error(ctx));       \
            return 1;                                                                       \
        }                                                                                   \
    } while (0)

/* a frame with a few bright rectangles on a ramp: enough structure for all three detectors */
static void draw(uint8_t *p, int nx, int ny, int f)
{
    for (int y = 0; y < ny; y++)
        for (int x = 0; x < nx; x++) p[(size_t)y * nx + x] = (uint8_t)(((x + y) >> 3) & 31);
    for (int k = 0; k < 12; k++) {
        const int x0 = (37 * k + 11 * f) % (nx - 40), y0 = (53 * k + 7 * f) % (ny - 30);
        for (int y = y0; y < y0 + 20 + k; y++)
            for (int x = x0; x < x0 + 30 + k; x++) p[(size_t)y * nx + x] = (uint8_t)(90 + 12 * k);
    }
}

int main(int argc, char **argv)
{
    const int nx = argc > 1 ? atoi(argv[1]) : 640, ny = argc > 2 ? atoi(argv[2]) : 480, n = argc > 3 ? atoi(argv[3]) : 10;
    const int batch = 4;
    imgfd_ctx *ctx = NULL;
    if (nx < 64 || ny < 64 || n < 1 || imgfd_ctx_create(0, &ctx) != IMGFD_OK) {
        fprintf(stderr, "needs a gfx950 device and frames of at least 64x64\n");
        return 1;
    }
    imgfd_stream_params p;
    imgfd_stream_default_params(&p); /* the defaults of image_harris(), image_detect_corners(), image_canny_edge_detector() */
    p.corner_cap = 256;              /* keep up to 256 Harris corners per frame */
    imgfd_stream *st = NULL;
    CHECK(imgfd_stream_open(ctx, nx, ny, batch, &p, &st));
    const size_t fb = (size_t)nx * ny;
    uint8_t *buf[3];
    for (int i = 0; i < 3; i++)
        if (!(buf[i] = (uint8_t *)imgfd_host_alloc(fb * batch))) return 1; /* pinned: read by the DMA engine directly */
    imgfd_stream_result r;
    int submitted = 0, pending = 0, b = 0;
    while (submitted < n || pending) {
        if (submitted < n && pending < 2) {
            const int m = n - submitted < batch ? n - submitted : batch;
            for (int f = 0; f < m; f++) draw(buf[b % 3] + fb * f, nx, ny, submitted + f);
            CHECK(imgfd_stream_submit(st, buf[b % 3], m, fb));
            submitted += m; pending++; b++;
            continue;
        }
        CHECK(imgfd_stream_collect(st, &r));
        pending--;
        for (int f = 0; f < r.n_frames; f++) {
            printf("frame %lld: harris %lld fast9 %lld canny %lld", (long long)(r.first_frame + f), (long long)r.harris_counts[f],
                   (long long)r.fast9_counts[f], (long long)r.canny_counts[f]);
            if (r.harris_counts[f] > 0) {
                const imgfd_corner *c = r.corners + (size_t)f * p.corner_cap;
                printf("  first corner (%.0f, %.0f) R = %.1f", c->x, c->y, c->R);
            }
            printf("\n");
        }
    }
    imgfd_stream_close(st);
    for (int i = 0; i < 3; i++) imgfd_host_free(buf[i]);
    imgfd_ctx_destroy(ctx);
    return 0;
}
