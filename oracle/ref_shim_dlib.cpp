/*
 * oracle/ref_shim_dlib.cpp -- TEST INFRASTRUCTURE.
 * extern "C" doorways into the UNMODIFIED dlib 19.20 headers bundled with the reference
 * (image.dlib/inst/dlib-19.20), compiled in place by oracle/Makefile into oracle/_ref/libref_dlib.so
 * with the flags R CMD INSTALL uses (-O2, no -march: the SSE2 code paths of dlib/simd).
 * Marshalling follows the reference glue: image.dlib/src/rcpp_fhog.cpp:17-38, rcpp_surf.cpp:14-52.
 */
#include <dlib/pixel.h>
#include <dlib/array2d.h>
#include <dlib/matrix.h>
#include <dlib/image_transforms/fhog.h>
#include <dlib/image_keypoint.h>

#include <string.h>
#include <vector>

using namespace dlib;

static void load_rgb(array2d<rgb_pixel> &img, const unsigned char *rgb, int rows, int cols)
{
    img.set_size(rows, cols);
    for (int r = 0; r < rows; r++)
        for (int c = 0; c < cols; c++) {
            const unsigned char *p = rgb + (size_t)3 * (c + (size_t)r * cols);
            assign_pixel(img[r][c], rgb_pixel(p[0], p[1], p[2]));
        }
}

extern "C" {

/* extract_fhog_features; out: hog[r][c][31] (row-major AoS), may be NULL to query the size */
int ref_fhog(const unsigned char *rgb, int rows, int cols, int cell_size, int pad_r, int pad_c, float *out,
             int *hog_nr, int *hog_nc)
{
    array2d<rgb_pixel> img;
    load_rgb(img, rgb, rows, cols);
    array2d<matrix<float, 31, 1> > hog;
    extract_fhog_features(img, hog, cell_size, pad_r, pad_c);
    *hog_nr = (int)hog.nr();
    *hog_nc = (int)hog.nc();
    if (out)
        for (long r = 0; r < hog.nr(); r++)
            for (long c = 0; c < hog.nc(); c++)
                for (int f = 0; f < 31; f++) out[((size_t)r * hog.nc() + c) * 31 + f] = hog[r][c](f);
    return 0;
}

/* integral_image_generic<int32>::load + raw table (rows x cols int32, inclusive prefix sums) */
int ref_integral(const unsigned char *rgb, int rows, int cols, int *out)
{
    array2d<rgb_pixel> img;
    load_rgb(img, rgb, rows, cols);
    integral_image_generic<int> ii;
    ii.load(img);
    for (long r = 0; r < rows; r++)
        for (long c = 0; c < cols; c++) out[(size_t)r * cols + c] = ii.get_sum_of_area(rectangle(0, 0, c, r));
    return 0;
}

/* hessian_pyramid::build_pyramid(int_img, 4, 6, 2) and get_interest_points: records of 6 doubles
 * (x, y, scale, score, laplacian, 0) in the order get_interest_points emits them */
long ref_surf_interest_points(const unsigned char *rgb, int rows, int cols, double threshold, double *out, long cap)
{
    array2d<rgb_pixel> img;
    load_rgb(img, rgb, rows, cols);
    integral_image_generic<int> ii;
    ii.load(img);
    hessian_pyramid pyr;
    pyr.build_pyramid(ii, 4, 6, 2);
    std::vector<interest_point> points;
    get_interest_points(pyr, threshold, points);
    for (size_t i = 0; i < points.size() && (long)i < cap; i++) {
        double *o = out + 6 * i;
        o[0] = points[i].center(0); o[1] = points[i].center(1); o[2] = points[i].scale;
        o[3] = points[i].score; o[4] = points[i].laplacian; o[5] = 0;
    }
    return (long)points.size();
}

/* one pyramid level: det-of-hessian values (with sign of the laplacian packed in) for octave o, interval i */
int ref_surf_pyramid_level(const unsigned char *rgb, int rows, int cols, int o, int i, double *out, int *nr, int *nc,
                           int *border)
{
    array2d<rgb_pixel> img;
    load_rgb(img, rgb, rows, cols);
    integral_image_generic<int> ii;
    ii.load(img);
    hessian_pyramid pyr;
    pyr.build_pyramid(ii, 4, 6, 2);
    *nr = (int)pyr.nr(o);
    *nc = (int)pyr.nc(o);
    *border = (int)pyr.get_border_size(i);
    if (out) {
        const int b = *border;
        for (long r = 0; r < *nr; r++)
            for (long c = 0; c < *nc; c++) {
                /* cells outside [border, n-border) are uninitialised in the reference: report 0 there */
                const bool inside = r >= b && r < *nr - b && c >= b && c < *nc - b;
                out[(size_t)r * *nc + c] = inside ? pyr.get_value(o, i, r, c) * (pyr.get_laplacian(o, i, r, c) < 0 ? -1.0 : 1.0) : 0.0;
            }
    }
    return 0;
}

/* get_surf_points: records of 71 doubles (x, y, angle, scale, score, laplacian, 0, des[64]) */
long ref_surf(const unsigned char *rgb, int rows, int cols, long max_points, double threshold, double *out, long cap)
{
    array2d<rgb_pixel> img;
    load_rgb(img, rgb, rows, cols);
    std::vector<surf_point> sp = get_surf_points(img, max_points, threshold);
    for (size_t i = 0; i < sp.size() && (long)i < cap; i++) {
        double *o = out + 71 * i;
        o[0] = sp[i].p.center(0); o[1] = sp[i].p.center(1); o[2] = sp[i].angle; o[3] = sp[i].p.scale;
        o[4] = sp[i].p.score; o[5] = sp[i].p.laplacian; o[6] = 0;
        for (int j = 0; j < 64; j++) o[7 + j] = sp[i].des(j);
    }
    return (long)sp.size();
}

}  // extern "C"
