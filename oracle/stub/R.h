/* oracle/stub/R.h -- build-time stand-in for R's <R.h> so the reference Harris
 * sources (which only need Rprintf, harris.cpp:8) compile without an R install.
 * Test infrastructure only. */
#pragma once
#include <stdio.h>
#define Rprintf printf
