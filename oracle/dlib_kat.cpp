/*
 * oracle/dlib_kat.cpp -- TEST INFRASTRUCTURE.
 * Dumps dlib's OWN known-answer vectors for extract_fhog_features -- the base64/compressed `face.dng` image and the
 * serialized feature arrays embedded in image.dlib/inst/dlib-19.20/dlib/test/fhog.cpp:156-214 (face_dng :215,
 * fhog_feats :347, fhog_grayscale :585) -- into one flat binary file that scripts/make_golden.py turns into
 * tests/golden/fhog_dlib_kat.npz.  The test file is #included where it lies (nothing is copied); its private string
 * getters are reached by compiling this translation unit with `private` spelled `public` around that include.
 * Built by `make -C oracle dlib_kat` (needs /root/reference; ~1 min: it links dlib/all/source.cpp for base64,
 * compress_stream and the test harness).
 *
 * File layout (little endian): int32 rows, cols; rows*cols*3 u8 RGB; rows*cols u8 gray (dlib's assign_image);
 * then three records { int32 cell_size, nr, nc; nr*nc*31 float32 in array2d order }: RGB sbin1, RGB sbin2, gray.
 */
#include <dlib/image_transforms.h>
#include <dlib/compress_stream.h>
#include <dlib/base64.h>
#include <dlib/image_io.h>

#include <cstdio>
#include <fstream>
#include <sstream>
#include <vector>

#include "tester.h"
#define private public
#include "fhog.cpp"  /* -I<dlib>/dlib/test */
#undef private

static void put_i32(std::ofstream &f, int v) { f.write((const char *)&v, 4); }
static void put_hog(std::ofstream &f, int sbin, const dlib::array2d<dlib::matrix<float, 31, 1> > &h)
{
    put_i32(f, sbin); put_i32(f, (int)h.nr()); put_i32(f, (int)h.nc());
    for (long r = 0; r < h.nr(); r++)
        for (long c = 0; c < h.nc(); c++)
            for (int o = 0; o < 31; o++) { const float v = h[r][c](o); f.write((const char *)&v, 4); }
}

int main(int argc, char **argv)
{
    if (argc < 2) { fprintf(stderr, "usage: dlib_kat out.bin\n"); return 2; }
    using namespace dlib;
    array2d<rgb_pixel> img;
    array2d<unsigned char> gimg;
    std::istringstream sin(a.get_decoded_string_face_dng());   /* fhog.cpp:167-169 */
    load_dng(img, sin);
    assign_image(gimg, img);
    int sbin1, sbin2, gsbin1;
    array2d<matrix<float, 31, 1> > vhog1, vhog2, gvhog1;
    sin.clear(); sin.str(a.get_decoded_string_fhog_feats());     /* fhog.cpp:171-177 */
    deserialize(sbin1, sin); deserialize(vhog1, sin); deserialize(sbin2, sin); deserialize(vhog2, sin);
    sin.clear(); sin.str(a.get_decoded_string_fhog_grayscale()); /* fhog.cpp:179-181 */
    deserialize(gsbin1, sin); deserialize(gvhog1, sin);
    std::ofstream f(argv[1], std::ios::binary);
    put_i32(f, (int)img.nr()); put_i32(f, (int)img.nc());
    for (long r = 0; r < img.nr(); r++)
        for (long c = 0; c < img.nc(); c++) { const unsigned char p[3] = {img[r][c].red, img[r][c].green, img[r][c].blue}; f.write((const char *)p, 3); }
    for (long r = 0; r < gimg.nr(); r++) f.write((const char *)&gimg[r][0], gimg.nc());
    put_hog(f, sbin1, vhog1); put_hog(f, sbin2, vhog2); put_hog(f, gsbin1, gvhog1);
    /* dlib's own criterion on its own implementation, as a self-check of this extraction (fhog.cpp:33-52) */
    array2d<matrix<float, 31, 1> > hog;
    extract_fhog_features(img, hog, sbin1);
    double worst = 0;
    for (long r = 0; r < hog.nr(); r++)
        for (long c = 0; c < hog.nc(); c++) worst = std::max<double>(worst, max(abs(hog[r][c] - vhog1[r][c])));
    printf("face.dng %ldx%ld, sbin %d/%d/%d, hog %ldx%ld, dlib-vs-golden max abs err %.3g\n", img.nr(), img.nc(), sbin1, sbin2, gsbin1,
           vhog1.nr(), vhog1.nc(), worst);
    return worst < 1e-6 ? 0 : 1;
}
