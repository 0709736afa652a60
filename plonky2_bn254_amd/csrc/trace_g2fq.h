// Trace-generation drivers of the G2 scalar-mul and Fq-exp STARKs (trace_g2fq.hip).
#pragma once
#include "gl_dev.h"
#include "layout.h"
size_t g2_trace_scratch_words(size_t n);
size_t fq_trace_scratch_words(size_t n);
// device pointers; trace column-major [W][N]; outputs n x 16 (G2) / n x 4 (Fq) canonical words
int g2_generate_trace_device(const u64* d_scalars, const u64* d_x, const u64* d_off, size_t n, u64* d_trace, size_t N,
                             u64* d_scratch, u64* d_outputs, int* d_err, hipStream_t st, bool with_range = true);
int fq_generate_trace_device(const u64* d_scalars, const u64* d_x, size_t n, u64* d_trace, size_t N, u64* d_scratch,
                             u64* d_outputs, int* d_err, hipStream_t st, bool with_range = true);
