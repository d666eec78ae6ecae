/* r/imgfd_glue.h -- shared by the four glue files: one imgfd context per R session and the error bridge.
 * R calls .Call from its single main thread; Rf_error is raised only here, after the library call has returned
 * (nothing throws or longjmps across the C ABI; reference behaviour: BEGIN_RCPP/END_RCPP, RcppExports.cpp). */
#ifndef IMGFD_GLUE_H
#define IMGFD_GLUE_H
#include <Rinternals.h>
#include <stdint.h>
#include "imgfd.h"

static imgfd_ctx *imgfd_glue_ctx_slot = NULL; /* one context per package DLL, created on first use */
static imgfd_ctx *imgfd_glue_ctx(void)
{
    if (!imgfd_glue_ctx_slot && imgfd_ctx_create(0, &imgfd_glue_ctx_slot) != IMGFD_OK) {
        imgfd_glue_ctx_slot = NULL;
        Rf_error("imgfd: no MI355X device / HIP runtime");
    }
    return imgfd_glue_ctx_slot;
}
/* R_unload_<pkg>: give the stream, the device workspace and the pinned buffers back when the package's DLL is unloaded */
static void imgfd_glue_unload(void)
{
    if (imgfd_glue_ctx_slot) imgfd_ctx_destroy(imgfd_glue_ctx_slot);
    imgfd_glue_ctx_slot = NULL;
}
static void imgfd_glue_check(imgfd_status s)
{
    if (s != IMGFD_OK) Rf_error("imgfd: %s", imgfd_last_error(imgfd_glue_ctx()));
}
#endif
