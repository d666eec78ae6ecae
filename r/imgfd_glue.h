/* r/imgfd_glue.h -- shared by the four glue files: one imgfd context per R session and the error bridge.
 * R calls .Call from its single main thread; Rf_error is raised only here, after the library call has returned
 * (nothing throws or longjmps across the C ABI; reference behaviour: BEGIN_RCPP/END_RCPP, RcppExports.cpp). */
#ifndef IMGFD_GLUE_H
#define IMGFD_GLUE_H
#include <Rinternals.h>
#include <stdint.h>
#include "imgfd.h"

static imgfd_ctx *imgfd_glue_ctx(void)
{
    static imgfd_ctx *c = NULL;
    if (!c && imgfd_ctx_create(0, &c) != IMGFD_OK) Rf_error("imgfd: no MI355X device / HIP runtime");
    return c;
}
static void imgfd_glue_check(imgfd_status s)
{
    if (s != IMGFD_OK) Rf_error("imgfd: %s", imgfd_last_error(imgfd_glue_ctx()));
}
#endif
