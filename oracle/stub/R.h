/* oracle/stub/R.h -- build-time stand-in for R's <R.h> so the reference sources compile without an R install:
 * Harris needs Rprintf (harris.cpp:8), Canny's tools.c needs Rf_error (tools.c:39-42).  Test infrastructure only. */
#pragma once
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#define Rprintf printf
static inline void Rf_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vfprintf(stderr, fmt, ap);
    va_end(ap);
    fputc('\n', stderr);
    abort();
}
