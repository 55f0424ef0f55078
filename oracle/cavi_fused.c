/*
 * oracle/cavi_fused.c -- TEST INFRASTRUCTURE / CPU BASELINE, NOT PRODUCT CODE.
 * The fused OpenMP comparator (SURVEY.md 8(d) variant ii); see cavi_fused_impl.h.
 * Only tests/ and bench.py's cpu_baseline leg may load it.
 */
#include <math.h>
#include <omp.h>
#include <stdlib.h>

double orc_psi(double x);   /* cavi_oracle.c */

#define REAL double
#define SUFFIX _f64
#define RLOG log
#define REXP exp
#include "cavi_fused_impl.h"
#undef REAL
#undef SUFFIX
#undef RLOG
#undef REXP

#define REAL float
#define SUFFIX _f32
#define RLOG logf
#define REXP expf
#include "cavi_fused_impl.h"
#undef REAL
#undef SUFFIX
#undef RLOG
#undef REXP
