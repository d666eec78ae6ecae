/* examples/multi_gpu_counts.c -- the path's multi-GPU form from plain C: frames are independent, so a node's GPUs each take
 * a contiguous block of the stream -- one imgfd context and one frame stream per device, one host thread
 * per device, no exchange between devices; the only "collective" is the sum of the per-device feature counts, done here
 * on the host (bench.py does it with RCCL on device tensors).
 *
 *   gcc -std=c99 -O2 -pthread -Iinclude examples/multi_gpu_counts.c -o multi_gpu_counts image_amd/libimgfd.so -Wl,-rpath,$PWD/image_amd
 *   ./multi_gpu_counts [nx ny n_frames [n_devices]]
 */
#define _POSIX_C_SOURCE 200809L
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "imgfd.h"

typedef struct {
    int device, nx, ny, first, count;  /* this device's block of the stream: frames first .. first + count - 1 */
    long long harris, fast9, canny;    /* feature totals of the block */
    double seconds;                    /* this device's block, first submit to last collect (frame drawing included) */
    int status;
    char err[256];
} shard;

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

static int fail(shard *s, imgfd_ctx *ctx, const char *what, imgfd_status st)
{
    snprintf(s->err, sizeof s->err, "device %d: %s -> %d: %s", s->device, what, (int)st, ctx ? imgfd_last_error(ctx) : "");
    s->status = (int)st;
    return 1;
}

static void *run(void *arg)
{
    shard *s = (shard *)arg;
    const int batch = 4;
    imgfd_ctx *ctx = NULL;
    imgfd_stream *st = NULL;
    imgfd_status rc = imgfd_ctx_create(s->device, &ctx);
    if (rc != IMGFD_OK) { fail(s, NULL, "imgfd_ctx_create", rc); return NULL; }
    imgfd_stream_params p;
    imgfd_stream_default_params(&p);
    if ((rc = imgfd_stream_open(ctx, s->nx, s->ny, batch, &p, &st)) != IMGFD_OK) { fail(s, ctx, "imgfd_stream_open", rc); imgfd_ctx_destroy(ctx); return NULL; }
    const size_t fb = (size_t)s->nx * s->ny;
    uint8_t *buf[3] = {NULL, NULL, NULL};
    for (int i = 0; i < 3; i++)
        if (!(buf[i] = (uint8_t *)imgfd_host_alloc(fb * batch))) { fail(s, ctx, "imgfd_host_alloc", IMGFD_ERR_OOM); goto done; }
    {
        imgfd_stream_result r;
        int submitted = 0, pending = 0, b = 0;
        struct timespec t0, t1;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        while (submitted < s->count || pending) {
            if (submitted < s->count && pending < 2) {
                const int m = s->count - submitted < batch ? s->count - submitted : batch;
                for (int f = 0; f < m; f++) draw(buf[b % 3] + fb * f, s->nx, s->ny, s->first + submitted + f);
                if ((rc = imgfd_stream_submit(st, buf[b % 3], m, fb)) != IMGFD_OK) { fail(s, ctx, "imgfd_stream_submit", rc); goto done; }
                submitted += m; pending++; b++;
                continue;
            }
            if ((rc = imgfd_stream_collect(st, &r)) != IMGFD_OK) { fail(s, ctx, "imgfd_stream_collect", rc); goto done; }
            pending--;
            for (int f = 0; f < r.n_frames; f++) {
                s->harris += r.harris_counts[f]; s->fast9 += r.fast9_counts[f]; s->canny += r.canny_counts[f];
            }
        }
        clock_gettime(CLOCK_MONOTONIC, &t1);
        s->seconds = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
    }
done:
    for (int i = 0; i < 3; i++) if (buf[i]) imgfd_host_free(buf[i]);
    imgfd_stream_close(st);
    imgfd_ctx_destroy(ctx);
    return NULL;
}

int main(int argc, char **argv)
{
    const int nx = argc > 1 ? atoi(argv[1]) : 640, ny = argc > 2 ? atoi(argv[2]) : 480, n = argc > 3 ? atoi(argv[3]) : 32;
    int ndev = 0;
    if (nx < 64 || ny < 64 || n < 1 || imgfd_device_count(&ndev) != IMGFD_OK) {
        fprintf(stderr, "needs a gfx950 device and frames of at least 64x64\n");
        return 1;
    }
    if (argc > 4 && atoi(argv[4]) >= 1 && atoi(argv[4]) < ndev) ndev = atoi(argv[4]);
    if (ndev > n) ndev = n;
    shard *sh = (shard *)calloc((size_t)ndev, sizeof *sh);
    pthread_t *th = (pthread_t *)calloc((size_t)ndev, sizeof *th);
    if (!sh || !th) return 1;
    for (int d = 0; d < ndev; d++) {  /* contiguous blocks, the first n % ndev devices take one frame more */
        const int q = n / ndev, rem = n % ndev;
        sh[d].device = d; sh[d].nx = nx; sh[d].ny = ny;
        sh[d].first = d * q + (d < rem ? d : rem);
        sh[d].count = q + (d < rem ? 1 : 0);
        if (pthread_create(&th[d], NULL, run, &sh[d])) return 1;
    }
    long long h = 0, f9 = 0, c = 0;
    int bad = 0;
    for (int d = 0; d < ndev; d++) {
        pthread_join(th[d], NULL);
        if (sh[d].status) { fprintf(stderr, "%s\n", sh[d].err); bad = 1; }
        printf("device %d: frames %d..%d: harris %lld fast9 %lld canny %lld  (%.3f s, %.1f Mpixel/s incl. drawing and upload)\n", d, sh[d].first,
               sh[d].first + sh[d].count - 1, sh[d].harris, sh[d].fast9, sh[d].canny, sh[d].seconds,
               sh[d].seconds > 0 ? 1e-6 * (double)sh[d].count * nx * ny / sh[d].seconds : 0.0);
        h += sh[d].harris; f9 += sh[d].fast9; c += sh[d].canny;
    }
    printf("total over %d device(s), %d frames: harris %lld fast9 %lld canny %lld\n", ndev, n, h, f9, c);
    free(sh); free(th);
    return bad;
}
