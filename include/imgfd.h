/*
 * imgfd.h -- C ABI of libimgfd.so, the MI355X (gfx950) feature-detection backend that replaces the
 * bundled CPU C/C++ behind five R functions of bnosac/image.  Plain pointers and sizes only.
 *
 * Reference interfaces replaced (paths relative to the bnosac/image repository):
 *   imgfd_harris   <- detect_corners(), image.CornerDetectionHarris/src/rcpp_harris.cpp:19-60, the body
 *                     of .Call symbol _image_CornerDetectionHarris_detect_corners (src/RcppExports.cpp:9-33)
 *   imgfd_fast9    <- detect_corners(), image.CornerDetectionF9/src/f9_rcpp.cpp:8-35
 *                     (_image_CornerDetectionF9_detect_corners, src/RcppExports.cpp:9-23); same argument
 *                     meaning as the library's own C API f9_detect_corners(), src/f9.h:77-86
 *   imgfd_canny    <- canny_edge_detector(), image.CannyEdges/src/rcpp_canny.cpp:122-244
 *                     (_image_CannyEdges_canny_edge_detector, src/RcppExports.cpp:9-24)
 *   imgfd_fhog     <- dlib_fhog(), image.dlib/src/rcpp_fhog.cpp:10-46 (_image_dlib_dlib_fhog)
 *   imgfd_surf     <- dlib_surf_points(), image.dlib/src/rcpp_surf.cpp:10-53 (_image_dlib_dlib_surf_points)
 * INTEGRATION.md shows the .Call glue that binds these from the unchanged R wrappers.
 *
 * Conventions
 *   - every function returns an imgfd_status (0 = ok); nothing throws or longjmps across this boundary;
 *     imgfd_last_error(ctx) gives the message of the last failure on that context.
 *   - a context owns one HIP stream, a grow-only device workspace and pinned staging buffers; it is
 *     thread-compatible (one thread at a time per context), like R's single-threaded .Call.
 *   - "host" entry points take host pointers (the drop-in path: the R glue hands over R-owned vectors);
 *     "_dev" entry points take DEVICE pointers to frames already resident in HBM and leave their results
 *     in device memory -- the batch/stream path the reference has no equivalent of.
 *   - images are row-major, index = x + nx*y (x fastest), exactly the buffers the reference's glue
 *     builds from the R vectors.
 */
#ifndef IMGFD_H
#define IMGFD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IMGFD_VERSION 1

#if defined(IMGFD_BUILD)
#define IMGFD_API __attribute__((visibility("default")))
#else
#define IMGFD_API
#endif

typedef enum {
    IMGFD_OK = 0,
    IMGFD_ERR_INVALID = 1,   /* bad argument; also a frame of more than 2^31 - 2^24 pixels (RGB: bytes): the kernels index one
                              * frame with 32 bits, as the reference's own int loops do (an R matrix holds < 2^31 elements) */
    IMGFD_ERR_NO_DEVICE = 2, /* no usable gfx950 device / HIP runtime failure at init */
    IMGFD_ERR_HIP = 3,       /* a HIP call failed; see imgfd_last_error */
    IMGFD_ERR_OOM = 4,
    IMGFD_ERR_UNSUPPORTED = 5
} imgfd_status;

typedef struct imgfd_ctx imgfd_ctx;

/* enum codes of the reference, passed through unchanged */
/* gaussian.h:14-16 */  enum { IMGFD_STD_GAUSSIAN = 0, IMGFD_FAST_GAUSSIAN = 1, IMGFD_NO_GAUSSIAN = 2 };
/* gradient.h:14-15 */  enum { IMGFD_CENTRAL_DIFFERENCES = 0, IMGFD_SOBEL_OPERATOR = 1 };
/* harris.h:17-19 */    enum { IMGFD_HARRIS_MEASURE = 0, IMGFD_SHI_TOMASI_MEASURE = 1, IMGFD_HARMONIC_MEAN_MEASURE = 2 };
/* harris.h:22-25 */    enum { IMGFD_ALL_CORNERS = 0, IMGFD_ALL_CORNERS_SORTED = 1, IMGFD_N_CORNERS = 2, IMGFD_DISTRIBUTED_N_CORNERS = 3 };
/* interpolation.h:13-15 */ enum { IMGFD_NO_INTERPOLATION = 0, IMGFD_QUADRATIC_APPROXIMATION = 1, IMGFD_QUARTIC_INTERPOLATION = 2 };

/* ------------------------------------------------------------------ context */
IMGFD_API imgfd_status imgfd_ctx_create(int device, imgfd_ctx **out);
/* same, but launch on an existing HIP stream (e.g. the caller's torch stream); stream is a hipStream_t */
IMGFD_API imgfd_status imgfd_ctx_create_on_stream(int device, void *stream, imgfd_ctx **out);
IMGFD_API void imgfd_ctx_destroy(imgfd_ctx *ctx);
IMGFD_API const char *imgfd_last_error(const imgfd_ctx *ctx);
/* hipStream_t of the context */
IMGFD_API void *imgfd_ctx_stream(imgfd_ctx *ctx);
IMGFD_API imgfd_status imgfd_ctx_sync(imgfd_ctx *ctx);
/* free a buffer returned through an out-parameter of this library */
IMGFD_API void imgfd_free(void *p);
IMGFD_API int imgfd_version(void);
/* number of HIP devices this process sees (0 and IMGFD_ERR_NO_DEVICE when there is none).  The path shards by frame: one
 * context per device, each with its own share of the frames, no exchange between them (examples/multi_gpu_counts.c). */
IMGFD_API imgfd_status imgfd_device_count(int *count);
/* arithmetic mode of the double-accumulated FIR (Harris Gaussians):
 *   0 = strict: the reference's operation sequence, no FMA anywhere (bit-exact planes)
 *   1 = fused-accumulate (default): sum = fma(B[j], pair, sum) inside the f64 accumulation only; the
 *       f32 stages (products, response) never contract.  Results differ from strict only when the
 *       f64 sum sits within ~1e-16 relative of a float rounding boundary. */
IMGFD_API imgfd_status imgfd_set_fir_mode(imgfd_ctx *ctx, int mode);
/* Switches: twelve, none changes a result.  Each selects a required fallback or is a knob the tests turn to reach a code path on small
 * inputs; the defaults are the measured best.  Each can also be given through the environment variable in brackets, which is read ONCE,
 * when the context is created.  (The measured-and-lost alternatives earlier rounds kept selectable -- sweep tile shapes, release points
 * and stream assignment of imgfd_detect_dev, hipGraph replay, tiled Harris NMS, plain-table SURF gathers, ... -- left the library in round 6:
 * LOG.md, scripts/experiments/r06_pruned_switches.patch.)  In brackets behind each: the GPU test that exercises it.
 *   "fir_mode" [IMGFD_FIR_MODE]  as imgfd_set_fir_mode  (tests/test_full_size.py::test_harris_4k, every strict-mode stage test)
 *   "fhog_fused" [IMGFD_FHOG_FUSED]  1 (default): cell_size 8 runs the fused gradient + histogram kernel; 0: the stage kernels that serve every
 *                  other cell size and width  (tests/test_fhog.py::test_fused_kernel_shapes)
 *   "fhog_bands" [IMGFD_FHOG_BANDS]  bands of 8 cell rows one workgroup of that kernel marches through (0: from the batch)  (same test)
 *   "hyst_sweeps" [IMGFD_HYST_SWEEPS]  Canny hysteresis: sweep launches queued before the union-find step (0: 9 for batches; up to 12 frames as
 *                  many as the recent calls on the context needed + 1, 8 at first)  (tests/test_canny.py::test_many_unconverged_frames_in_one_batch)
 *   "gauss_march" [IMGFD_GAUSS_MARCH]  1 (default): u8 frames whose width is a multiple of 16 (>= 256) take the marching Gaussian + gradient
 *                  kernel; 0: the tile kernel that serves every other shape  (tests/test_harris_stages.py::test_marching_gauss_grad_random_shapes)
 *   "gauss_march_seg" [IMGFD_GAUSS_MARCH_SEG]  rows per segment of the marching kernel (0: from the batch)  (same test)
 *   "tensor_workers" [IMGFD_TENSOR_WORKERS]  workgroups of fir_tensor, each with an equal share of the batch's line of 16-row chunk units
 *                  (0: one per compute unit)  (tests/test_harris_stages.py: the structure-tensor tests with few workers)
 *   "max_chunk_frames" [IMGFD_MAX_CHUNK_FRAMES]  frames per sub-batch of the *_dev entry points (0: from the memory budgets)  (tests/test_sub_batches.py)
 *   "surf_group" [IMGFD_SURF_GROUP]  4 (default): imgfd_surf_dev handles the tiles in groups of this many (1..16): a buffer set per tile,
 *                  the latency-bound back stages (maximum test, ranking, orientation, descriptor) as one launch each per group
 *   "surf_lanes" [IMGFD_SURF_LANES]  3 (default): the front stages (integral image, Hessian pyramid) of a group's tiles go round-robin
 *                  over this many HIP streams (1..4)  (both: tests/test_surf.py::test_surf_dev_groups_of_tiles)
 *   "surf_sort_cap" [IMGFD_SURF_SORT_CAP]  selected records imgfd_surf_dev ranks with its LDS sort (2048); more: all-pairs ranking
 *                  (tests/test_surf.py::test_surf_dev_ranks_and_cuts_on_the_device)
 *   "surf_rec_cap" [IMGFD_SURF_REC_CAP]  candidate records per tile imgfd_surf_dev buffers before the tile reports -needed (262144)
 *                  (tests/test_surf.py::test_surf_dev_redoes_tiles_whose_candidates_overflow)
 * Unknown names give IMGFD_ERR_INVALID. */
IMGFD_API imgfd_status imgfd_set_tuning(imgfd_ctx *ctx, const char *name, int value);
/* reads a switch back, or a statistic: "gauss_march_launches" (launches of the marching Gaussian + gradient kernel),
 * "canny_sweeps_queued" (sweep launches the last Canny call on this context queued), "canny_sweeps_working" /
 * "canny_frames_unconverged" (last Canny call on this context: the last sweep launch that changed a frame; frames the queued
 * launches did not finish -- these two wait for the stream) */
IMGFD_API imgfd_status imgfd_get_counter(imgfd_ctx *ctx, const char *name, int64_t *value);

/* ------------------------------------------------------------------ Harris */
typedef struct {
    float x, y; /* corner position (harris.h:33-46) */
    float R;    /* corner strength */
} imgfd_corner;

typedef struct {
    imgfd_corner *corners; /* library-allocated (imgfd_free), NULL when n == 0 */
    int64_t n;
    /* stage wall times in seconds, filled when verbose != 0: smooth, gradient, autocorrelation,
     * response, nms, select, subpixel (the seven lines harris.cpp:510-538 prints) */
    double stage_seconds[7];
} imgfd_corners;

/* Arguments exactly as rcpp_harris.cpp:19-32 after the NumericVector -> float narrowing
 * (rcpp_harris.cpp:34-35, which the glue performs): img holds nx*ny floats.  Silent-empty cases of the
 * reference are reproduced (nx<3 || ny<3: harris.cpp:493; image <= 2r+1: harris.cpp:151). */
IMGFD_API imgfd_status imgfd_harris(imgfd_ctx *ctx, const float *img, int nx, int ny, float k, float sigma_d,
                          float sigma_i, float threshold, int gaussian, int gradient, int strategy,
                          int Nselect, int measure, int Nscales, int precision, int cells, int verbose,
                          imgfd_corners *out);

/* ------------------------------------------------------------------ FAST-9 */
typedef struct { int x, y; } imgfd_point; /* F9_CORNER, f9.h:40-43 */

typedef struct {
    imgfd_point *points; /* library-allocated (imgfd_free); raster order like f9.cpp:2959-2960 */
    int64_t n;
} imgfd_points;

/* f9_detect_corners(ctx, image_data, width, height, bytes_per_row, threshold, suppress_non_max, &n),
 * f9.h:77-86.  Coordinates are the library's (x,y); the R-facing remap x<-y, y<-width-x of
 * f9_rcpp.cpp:29-30 belongs to the glue. */
IMGFD_API imgfd_status imgfd_fast9(imgfd_ctx *ctx, const uint8_t *img, int width, int height, int bytes_per_row,
                         uint8_t threshold, int suppress_non_max, imgfd_points *out);

/* ------------------------------------------------------------------ Canny */
/* canny_edge_detector(image, X, Y, s, low_thr, high_thr, accGrad), rcpp_canny.cpp:122-127, after the
 * IntegerVector -> unsigned char narrowing (:135-136).  edges: caller-allocated nx*ny bytes, 0 or 255
 * (:210-215); *pixels_nonzero as :226-233. */
IMGFD_API imgfd_status imgfd_canny(imgfd_ctx *ctx, const uint8_t *img, int nx, int ny, double s, double low_thr,
                         double high_thr, int accGrad, uint8_t *edges, int64_t *pixels_nonzero);

/* ------------------------------------------------------------------ fHOG (dlib)
 * dlib_fhog(x, rows, cols, cell_size, filter_rows_padding, filter_cols_padding), rcpp_fhog.cpp:10-15, after the
 * std::vector<int> -> rgb_pixel narrowing (:17-24, performed by the glue): rgb holds rows*cols interlaced
 * (r,g,b) bytes, pixel (r,c) at 3*(c + r*cols).  *hog (library-allocated, imgfd_free) holds
 * 31 * hog_nc * hog_nr floats in the order the reference glue emits them (rcpp_fhog.cpp:29-38): feature-major, then
 * column x, then row y fastest -- R reshapes it to [hog_height, hog_width, 31] (image_fhog.R:46).  An image too small
 * for 3x3 cells gives *hog = NULL, 0 x 0 (hog.clear(), fhog.h:783-812).  cell_size == 1 takes dlib's special case
 * (fhog.h:499-694). */
IMGFD_API imgfd_status imgfd_fhog(imgfd_ctx *ctx, const uint8_t *rgb, int rows, int cols, int cell_size,
                        int filter_rows_padding, int filter_cols_padding, float **hog, int *hog_nr, int *hog_nc);
/* output size of imgfd_fhog / imgfd_fhog_dev for a rows x cols image (no device needed) */
IMGFD_API imgfd_status imgfd_fhog_size(int rows, int cols, int cell_size, int filter_rows_padding, int filter_cols_padding,
                             int *hog_nr, int *hog_nc);
/* batch of n_frames interlaced RGB frames resident in HBM (frame f at d_rgb + f*frame_stride_bytes, rows packed);
 * d_hog: n_frames * 31*hog_nc*hog_nr floats, each frame in the layout of imgfd_fhog */
IMGFD_API imgfd_status imgfd_fhog_dev(imgfd_ctx *ctx, const uint8_t *d_rgb, int n_frames, int rows, int cols,
                            size_t frame_stride_bytes, int cell_size, int filter_rows_padding, int filter_cols_padding,
                            float *d_hog);

/* ------------------------------------------------------------------ SURF (dlib)
 * dlib_surf_points(x, rows, cols, max_points, detection_threshold), rcpp_surf.cpp:10-13, after the std::vector<int> ->
 * rgb_pixel narrowing (:14-21): rgb as for imgfd_fhog.  Output = the vectors the reference glue builds (:23-52):
 * n points in get_surf_points' order (descending score), surf[i*64 + j] = descriptor j of point i.  One allocation:
 * release with imgfd_free(out->data).  (R's NaN -> 0 patch of image_surf.R:88 belongs to the R wrapper.) */
typedef struct {
    int64_t n;
    double *x, *y, *angle, *pyramid_scale, *score, *laplacian; /* n each */
    double *surf;                                              /* n * 64 */
    double *data;                                              /* the one allocation behind the pointers above */
} imgfd_surf_out;
IMGFD_API imgfd_status imgfd_surf(imgfd_ctx *ctx, const uint8_t *rgb, int rows, int cols, long max_points,
                        double detection_threshold, imgfd_surf_out *out);
/* device stages only (integral image, Hessian pyramid, 3x3x3 maxima + interpolation), hessian_pyramid.h:453-506:
 * records of 5 doubles (x, y, scale, score, laplacian) in the order get_interest_points() emits them; *n = number found
 * (may exceed cap; only cap are stored). */
IMGFD_API imgfd_status imgfd_surf_interest_points(imgfd_ctx *ctx, const uint8_t *rgb, int rows, int cols,
                                        double detection_threshold, double *points, int64_t cap, int64_t *n);
/* batch of n_frames interlaced RGB tiles resident in HBM: the device stages of imgfd_surf_interest_points per tile.
 * d_points: n_frames * cap records in NO particular order; sort by `key` (octave, interval, row, column packed in
 * ascending significance) to get the order the reference emits them in; d_counts[f] = points found in tile f (may
 * exceed cap; only cap are stored). */
typedef struct {
    uint64_t key; /* ((octave*8 + interval) << 40) | (row << 20) | column, in pyramid-level coordinates */
    double x, y, scale, score, laplacian;
} imgfd_surf_point;
IMGFD_API imgfd_status imgfd_surf_points_dev(imgfd_ctx *ctx, const uint8_t *d_rgb, int n_frames, int rows, int cols,
                                   size_t frame_stride_bytes, double detection_threshold, imgfd_surf_point *d_points,
                                   int64_t cap, int64_t *d_counts);
/* stage doorway: the int32 integral image (integral_image.h:33-62) of the (r+g+b)/3 gray image, rows*cols values */
/* Batch form with K19 (orientation + descriptor, surf.h:75-232) on the device as well: n_frames RGB tiles in HBM in,
 * finished SURF features in HBM out.  d_features holds n_frames*cap records of 70 doubles -- x, y, angle,
 * pyramid_scale, score, laplacian, surf[64]: the columns dlib_surf_points returns (rcpp_surf.cpp:31-52) -- strongest
 * first as get_surf_points orders them (surf.h:268-285); d_counts[f] = records of frame f (<= min(max_points, cap)).
 * atan2/sin/cos come from the device libm here, so angles and descriptors agree with imgfd_surf to ~1e-12 rather than
 * bit for bit (SURVEY.md 8d asks for 1e-6); the interest points themselves are identical.  No tile waits for the host:
 * the candidates of a tile are ranked on the device (the max_points strongest, strongest first; two candidates with
 * exactly equal scores keep the order get_interest_points emitted them in -- the reference leaves that order to std::sort)
 * and the descriptor kernels read the point count on the device.  The call only QUEUES work (on the context's stream and
 * streams of its own that the context's stream waits for at the end): it never waits for the device, so it is safe on a
 * caller's stream.  A tile with more candidates than the record buffer holds (262144: a very low detection_threshold on a
 * large tile) reports d_counts[f] = -(records it needs room for) and leaves its feature rows untouched; imgfd_surf_dev_redo, called with the
 * same arguments, waits for the context's stream, reads the counts of the batch back once and redoes such tiles with a buffer
 * of the size they asked for (*n_redone, optional: how many).  A caller that cannot see such tiles (the R default threshold 30
 * on a 4096^2 tile yields ~10^4 candidates) need not call it. */
IMGFD_API imgfd_status imgfd_surf_dev(imgfd_ctx *ctx, const uint8_t *d_rgb, int n_frames, int rows, int cols,
                                      size_t frame_stride_bytes, long max_points, double detection_threshold,
                                      double *d_features, int64_t cap, int64_t *d_counts);
IMGFD_API imgfd_status imgfd_surf_dev_redo(imgfd_ctx *ctx, const uint8_t *d_rgb, int n_frames, int rows, int cols,
                                           size_t frame_stride_bytes, long max_points, double detection_threshold,
                                           double *d_features, int64_t cap, int64_t *d_counts, int *n_redone);
IMGFD_API imgfd_status imgfd_k_surf_integral(imgfd_ctx *ctx, const uint8_t *rgb, int rows, int cols, int32_t *out);

/* ------------------------------------------------------------------ descriptor matching
 * Exact k nearest neighbours of every query row among the data rows, Euclidean distance: what the reference's README
 * does with two images' `surf` matrices, FNN::get.knnx(sp1$surf, sp2$surf, k = 1) (image.dlib/README.md:19-37; FNN is a
 * CRAN package outside the reference tree).  nn_index: n_query*k, row-major, 0-BASED data row (R glue adds 1), -1 when
 * fewer than k data rows exist; nn_dist: the distances, ascending; equal distances rank by ascending index.
 * 1 <= dim <= 64, 1 <= k <= 8.  column_major != 0: the matrices are R matrices (n rows, dim columns, column-major);
 * otherwise rows are contiguous (imgfd_surf_out.surf). */
IMGFD_API imgfd_status imgfd_knn(imgfd_ctx *ctx, const double *data, int64_t n_data, const double *query, int64_t n_query,
                                 int dim, int k, int column_major, int32_t *nn_index, double *nn_dist);
/* Device pointers; element (i, j) of a matrix is p[i*row_stride + j*col_stride] (strides in doubles), so the records
 * of imgfd_surf_dev can be matched in place (p = d_features + 6, row_stride 70, col_stride 1).  Asynchronous on the
 * context's stream. */
IMGFD_API imgfd_status imgfd_knn_dev(imgfd_ctx *ctx, const double *d_data, int64_t n_data, int64_t data_row_stride,
                                     int64_t data_col_stride, const double *d_query, int64_t n_query,
                                     int64_t query_row_stride, int64_t query_col_stride, int dim, int k,
                                     int32_t *d_nn_index, double *d_nn_dist);

/* ------------------------------------------------------------------ R-native vectors
 * The same entry points for the vectors R actually holds -- what REAL(x) / INTEGER(x) point to -- so the glue needs no
 * element-by-element narrowing loop on R's single thread: the vector crosses PCIe once and the reference's casts
 * ((float) x[i], rcpp_harris.cpp:34-35; (unsigned char) x[i], f9_rcpp.cpp:10-11, rcpp_canny.cpp:135-136;
 * rgb_pixel(x[i], ..), rcpp_fhog.cpp:17-24, rcpp_surf.cpp:14-21) are applied in HBM.  Arguments and results are
 * otherwise those of the functions above; bytes_per_row / the (3, W, H) interlacing count ELEMENTS here. */
IMGFD_API imgfd_status imgfd_harris_f64(imgfd_ctx *ctx, const double *x, int nx, int ny, float k, float sigma_d,
                              float sigma_i, float threshold, int gaussian, int gradient, int strategy,
                              int Nselect, int measure, int Nscales, int precision, int cells, int verbose,
                              imgfd_corners *out);
IMGFD_API imgfd_status imgfd_fast9_i32(imgfd_ctx *ctx, const int32_t *x, int width, int height, int bytes_per_row,
                             uint8_t threshold, int suppress_non_max, imgfd_points *out);
IMGFD_API imgfd_status imgfd_canny_i32(imgfd_ctx *ctx, const int32_t *image, int nx, int ny, double s, double low_thr,
                             double high_thr, int accGrad, uint8_t *edges, int64_t *pixels_nonzero);
IMGFD_API imgfd_status imgfd_fhog_i32(imgfd_ctx *ctx, const int32_t *x, int rows, int cols, int cell_size,
                            int filter_rows_padding, int filter_cols_padding, float **hog, int *hog_nr, int *hog_nc);
IMGFD_API imgfd_status imgfd_surf_i32(imgfd_ctx *ctx, const int32_t *x, int rows, int cols, long max_points,
                            double detection_threshold, imgfd_surf_out *out);
/* The way back, for the two functions whose result is a per-pixel / per-cell array: the reference glue widens it into R's
 * doubles element by element on the R thread (rcpp_canny.cpp:226-233: NumericMatrix edges(nx, ny), 8.3 M stores for a 4K
 * frame; rcpp_fhog.cpp:29-38).  These entry points widen in HBM and copy straight into the vector R allocated:
 *   imgfd_canny_f64out: edges = nx*ny doubles (0.0 / 255.0), e.g. REAL(Rf_allocMatrix(REALSXP, nx, ny));
 *   imgfd_fhog_f64out:  hog = hog_cap >= 31*hog_nr*hog_nc doubles (sizes from imgfd_fhog_size) in the order of
 *                       rcpp_fhog.cpp:29-38; an image too small for 3x3 cells gives *hog_nr = *hog_nc = 0. */
IMGFD_API imgfd_status imgfd_canny_f64out(imgfd_ctx *ctx, const int32_t *image, int nx, int ny, double s, double low_thr,
                                double high_thr, int accGrad, double *edges, int64_t *pixels_nonzero);
IMGFD_API imgfd_status imgfd_fhog_f64out(imgfd_ctx *ctx, const int32_t *x, int rows, int cols, int cell_size,
                               int filter_rows_padding, int filter_cols_padding, double *hog, int64_t hog_cap,
                               int *hog_nr, int *hog_nc);

/* ------------------------------------------------------------------ device-resident batch path
 * Frames live in HBM: frame f starts at (char*)d_frames + f*frame_stride_bytes, rows are row_stride
 * bytes apart.  Results stay on the device in caller-provided buffers so that a stream of frames can
 * be processed without host synchronisation; counts are int64 per frame.  All launches go to the
 * context's stream and return immediately. */
typedef struct {
    const void *d_frames;
    int n_frames;
    int nx, ny;
    size_t frame_stride_bytes;
    int row_stride_bytes;
    int dtype; /* 0 = u8, 1 = f32 (Harris only) */
} imgfd_frames;

/* Harris, reference default path and every enum code that runs on the device (gaussian 0/1/2,
 * gradient 0/1, measure 0/1/2); strategy "all corners", no sub-pixel, one scale: exactly what every
 * public image_harris() call executes.  d_corners: n_frames*cap records, raster order;
 * d_counts[f] = number of corners found in frame f (may exceed cap; only cap are stored).  cap = 0 asks for the
 * counts only; d_corners may then be NULL (likewise d_points of imgfd_fast9_dev). */
IMGFD_API imgfd_status imgfd_harris_dev(imgfd_ctx *ctx, const imgfd_frames *fr, float k, float sigma_d,
                              float sigma_i, float threshold, int gaussian, int gradient, int measure,
                              imgfd_corner *d_corners, int64_t cap, int64_t *d_counts);

IMGFD_API imgfd_status imgfd_fast9_dev(imgfd_ctx *ctx, const imgfd_frames *fr, uint8_t threshold,
                             int suppress_non_max, imgfd_point *d_points, int64_t cap,
                             int64_t *d_counts);

/* d_edges: n_frames * nx*ny bytes (0/255); d_counts[f] = pixels_nonzero of frame f */
IMGFD_API imgfd_status imgfd_canny_dev(imgfd_ctx *ctx, const imgfd_frames *fr, double s, double low_thr,
                             double high_thr, int accGrad, uint8_t *d_edges, int64_t *d_counts);

/* ------------------------------------------------------------------ stage doorways (parity + roofline)
 * Device pointers, one frame, planes of nx*ny floats.  These run the individual kernels of the Harris
 * path so tests can compare plane by plane and bench.py can time the structure-tensor pass alone. */
/* K1: discrete_gaussian / SII / copy, gaussian.cpp:403-430 (type as the reference codes) */
IMGFD_API imgfd_status imgfd_k_gaussian(imgfd_ctx *ctx, const float *d_in, float *d_out, int nx, int ny,
                              float sigma, int type);
/* K2: central differences / Sobel, gradient.cpp:115-128 */
IMGFD_API imgfd_status imgfd_k_gradient(imgfd_ctx *ctx, const float *d_I, float *d_Ix, float *d_Iy, int nx,
                              int ny, int type);
/* K1 + K2 as the batch path runs them on u8 frames: discrete Gaussian of radius 3 (sigma_d in [1, 4/3)) and the gradient of
 * the smoothed frame in one kernel (gaussian.cpp:289-395 + gradient.cpp:17-106); the smoothed plane is not written.
 * d_u8: ny rows of nx bytes (pitch nx).  IMGFD_ERR_UNSUPPORTED for another radius. */
IMGFD_API imgfd_status imgfd_k_gauss_grad_u8(imgfd_ctx *ctx, const uint8_t *d_u8, float *d_Ix, float *d_Iy, int nx, int ny,
                                             float sigma_d, int grad_type);
/* the gradient table of the fused fHOG kernel: 511 x 512 words, entry [(ty + 255) * 512 + tx + 255] for the integer gradient
 * (tx, ty): bits 0..26 = sqrtf(tx^2 + ty^2) with the exponent field lowered by 126 (0 for a zero gradient), bits 27..31 = the
 * orientation bin (fhog.h:846-859) */
IMGFD_API imgfd_status imgfd_k_fhog_lut(imgfd_ctx *ctx, uint32_t *d_out);
/* the same table computed without the float chain (integer orientation rule of fhog_fused.hip's fh_word_arith, what the
 * fused kernel evaluates for waves with many large gradients): equal to imgfd_k_fhog_lut's bit for bit */
IMGFD_API imgfd_status imgfd_k_fhog_lut_arith(imgfd_ctx *ctx, uint32_t *d_out);
/* K3: the structure-tensor pass, compute_autocorrelation_matrix harris.cpp:44-70:
 * reads Ix,Iy (8 B/px), writes the smoothed A,B,C (12 B/px) */
IMGFD_API imgfd_status imgfd_k_structure_tensor(imgfd_ctx *ctx, const float *d_Ix, const float *d_Iy, float *d_A,
                                      float *d_B, float *d_C, int nx, int ny, float sigma, int gauss);
/* K3 + K4 in one kernel, the form the batch path runs on image_harris() defaults: structure tensor and Harris corner
 * response (harris.cpp:44-70, :78-105), only R is written.  Discrete Gaussian with radius 7, 3 or 1 (sigma_i 2.5, 1.25,
 * 0.625), nx % 4 == 0, 16-byte aligned planes; IMGFD_ERR_UNSUPPORTED otherwise. */
IMGFD_API imgfd_status imgfd_k_tensor_response(imgfd_ctx *ctx, const float *d_Ix, const float *d_Iy, float *d_R, int nx,
                                               int ny, float sigma, float k);
/* K4: compute_corner_response harris.cpp:78-133 */
IMGFD_API imgfd_status imgfd_k_response(imgfd_ctx *ctx, const float *d_A, const float *d_B, const float *d_C,
                              float *d_R, int nx, int ny, int measure, float k);
/* K5: non_maximum_suppression harris.cpp:141-255 + raster-ordered compaction */
IMGFD_API imgfd_status imgfd_k_nms(imgfd_ctx *ctx, const float *d_R, int nx, int ny, float Th, int radius,
                         imgfd_corner *d_corners, int64_t cap, int64_t *d_count);

/* K5 as the batch path runs it: the threshold test of harris.cpp:160-162 as a byte per quad of pixels (there: written by
 * the structure-tensor kernel's epilogue), then non_maximum_suppression harris.cpp:141-255 reading R only around the
 * pixels that pass + raster-ordered compaction.  nx % 4 == 0; any radius. */
IMGFD_API imgfd_status imgfd_k_nms_quads(imgfd_ctx *ctx, const float *d_R, int nx, int ny, float Th, int radius,
                                         imgfd_corner *d_corners, int64_t cap, int64_t *d_count);

/* Time `iters` back-to-back launches of the structure-tensor kernel with HIP events on the context's
 * stream (after `warmup` untimed launches); *avg_us = mean microseconds per launch. */
IMGFD_API imgfd_status imgfd_time_structure_tensor(imgfd_ctx *ctx, const float *d_Ix, const float *d_Iy,
                                         float *d_A, float *d_B, float *d_C, int nx, int ny,
                                         float sigma, int gauss, int warmup, int iters, double *avg_us);

/* same over a batch of n_frames packed planes (frame f at p + f*nx*ny): one launch covers the batch */
IMGFD_API imgfd_status imgfd_time_structure_tensor_batch(imgfd_ctx *ctx, const float *d_Ix, const float *d_Iy,
                                               float *d_A, float *d_B, float *d_C, int nx, int ny, int n_frames,
                                               float sigma, int gauss, int warmup, int iters, double *avg_us);
/* human-readable name of the kernel imgfd_harris_dev launches for the structure-tensor pass on image_harris() defaults */
IMGFD_API const char *imgfd_tensor_kernel_name(imgfd_ctx *ctx);

/* In-pipeline timing of the structure-tensor pass: while enabled, every launch of that kernel made by
 * imgfd_harris / imgfd_harris_dev on this context is bracketed by HIP events on the context's stream.
 * imgfd_profile_k3_read synchronises, returns the summed device time and the number of launches since
 * the last read, and resets the counters. */
IMGFD_API imgfd_status imgfd_profile_k3(imgfd_ctx *ctx, int enable);
IMGFD_API imgfd_status imgfd_profile_k3_read(imgfd_ctx *ctx, double *total_us, int *launches);

/* The shader clock the device runs at while the library's kernels run (the chip clocks to its power budget: the same kernel
 * is 3-7 % slower on one box than on another, and slower beside an f64-heavy load than alone).  imgfd_clock_probe queues ONE
 * wavefront on a stream of the context's own -- beside whatever its other streams run, it does not wait for them and they do
 * not wait for it -- that reads the shader-cycle counter and the constant-rate wall counter span_us (1 .. 100000) microseconds
 * apart; imgfd_clock_probe_read waits for the probes queued since the last read and returns mean / min / max GHz over them
 * (samples = 0 and zeros when none was queued; at most 4096 probes are kept between two reads). */
IMGFD_API imgfd_status imgfd_clock_probe(imgfd_ctx *ctx, int span_us);
IMGFD_API imgfd_status imgfd_clock_probe_read(imgfd_ctx *ctx, double *mean_ghz, double *min_ghz, double *max_ghz, int *samples);

/* Fill n_frames synthetic u8 frames G(seed0+f) directly in HBM (image_amd/synth.py is the host twin). */
IMGFD_API imgfd_status imgfd_synth_frames(imgfd_ctx *ctx, uint8_t *d_frames, int n_frames, int nx, int ny,
                                size_t frame_stride_bytes, uint32_t seed0, const int32_t *d_rects,
                                int n_rect);

/* ------------------------------------------------------------------ host frame streams
 * A sequence of u8 gray frames that lives in HOST memory (decoded PGMs, a camera ring) run through Harris / FAST-9 /
 * Canny batch by batch: while batch i is in the kernels, batch i+1 crosses PCIe on a second HIP stream into the other
 * device buffer.  The reference has no counterpart -- image_harris(), image_detect_corners() and
 * image_canny_edge_detector() take one image per call (H/R/pkg.R:76-90, F9/R/image_detect_corners.R:10-27,
 * CE/R/canny_edges_detector.R:63) -- this is the driver loop a caller of those functions writes around them, moved below the boundary
 * so that the upload overlaps the compute.  Results of every frame are those of imgfd_harris_dev / imgfd_fast9_dev /
 * imgfd_canny_dev on the same frame.
 *
 * Frames handed to imgfd_stream_submit may be ordinary (pageable) memory; they are then staged through pinned
 * buffers by a helper thread.  Memory from imgfd_host_alloc (pinned) is read by the DMA engine directly. */
typedef struct imgfd_stream imgfd_stream;

typedef struct {
    int harris, fast9, canny;              /* which detectors run (non-zero = on) */
    float k, sigma_d, sigma_i, threshold;  /* Harris, as imgfd_harris_dev */
    int gaussian, gradient, measure;
    int fast9_threshold, suppress_non_max; /* FAST-9, as imgfd_fast9_dev */
    double s, low_thr, high_thr;           /* Canny, as imgfd_canny_dev */
    int accGrad;
    int64_t corner_cap, point_cap;         /* records kept per frame (0: counts only) */
    int keep_edges;                        /* non-zero: the 0/255 edge maps are copied back too */
} imgfd_stream_params;

typedef struct {
    int n_frames;                /* frames in this batch */
    int64_t first_frame;         /* index of its first frame in submission order */
    const int64_t *harris_counts, *fast9_counts, *canny_counts; /* n_frames each; NULL when the detector is off */
    const imgfd_corner *corners; /* frame f: corners + f*corner_cap, min(count, cap) records; NULL when cap is 0 */
    const imgfd_point *points;   /* frame f: points + f*point_cap */
    const uint8_t *edges;        /* n_frames * nx*ny bytes; NULL unless keep_edges */
} imgfd_stream_result;

/* One device-resident batch through the selected detectors, overlapped on two HIP streams (the Canny hysteresis
 * sweeps, which end in a few idle launches and terminate on the device, run beside FAST-9 and the Harris chain).  Results are
 * exactly those of imgfd_harris_dev / imgfd_fast9_dev / imgfd_canny_dev with the parameters in *p (keep_edges is
 * ignored: d_edges is required whenever Canny is on).  d_counts: 3*n_frames int64 -- Harris counts, then FAST-9, then
 * Canny pixels_nonzero; the third of a detector that is off is left untouched.  Work queued on the context's stream
 * before the call is waited for; after the call the context's stream (imgfd_ctx_sync) covers all results. */
IMGFD_API imgfd_status imgfd_detect_dev(imgfd_ctx *ctx, const imgfd_frames *fr, const imgfd_stream_params *p,
                                        imgfd_corner *d_corners, imgfd_point *d_points, uint8_t *d_edges,
                                        int64_t *d_counts);

/* defaults of the three R functions, all detectors on, counts only */
IMGFD_API void imgfd_stream_default_params(imgfd_stream_params *p);
IMGFD_API imgfd_status imgfd_stream_open(imgfd_ctx *ctx, int nx, int ny, int batch_frames,
                                         const imgfd_stream_params *params, imgfd_stream **out);
/* Start the upload of n_frames (1..batch_frames) frames, frame f at frames + f*frame_stride_bytes (rows contiguous),
 * and launch the kernels of the batch submitted before it.  Does not wait for either.  At most two batches can be
 * pending; a third submit without a collect fails with IMGFD_ERR_INVALID.  `frames` must stay valid until the
 * batch has been collected. */
IMGFD_API imgfd_status imgfd_stream_submit(imgfd_stream *st, const uint8_t *frames, int n_frames,
                                           size_t frame_stride_bytes);
/* Wait for the oldest pending batch and describe its results.  The arrays live in pinned memory owned by the stream
 * and stay valid until the second imgfd_stream_submit after this call.  res->n_frames == 0 when nothing is pending. */
IMGFD_API imgfd_status imgfd_stream_collect(imgfd_stream *st, imgfd_stream_result *res);
IMGFD_API void imgfd_stream_close(imgfd_stream *st);
/* pinned host memory for frames (hipHostMalloc); NULL on failure */
IMGFD_API void *imgfd_host_alloc(size_t bytes);
IMGFD_API void imgfd_host_free(void *p);

#ifdef __cplusplus
}
#endif
#endif /* IMGFD_H */
