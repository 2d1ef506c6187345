// lm_compile.h — the model compiler on the device: a freshly randomised model per environment (lm_compile.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace lmc {

// limits of the compiler kernel's LDS tables (lm_set_model_compiler refuses a program beyond them)
constexpr int kMaxNv = 36, kMaxDraw = 768, kMaxRbody = 64, kMaxGslot = 320, kMaxBody = 128, kMaxSlot = 32;
constexpr int kIntHead = 16, kDblHead = 8, kDrawInts = 4, kDrawDbls = 2, kRbInts = 4, kRbDbls = 55, kConInts = 8;
constexpr unsigned kMagic = 0x4C4D4D43u;   // "LMMC" (lowering.MC_MAGIC)

struct Args {
  const int* ib; const double* db;          // the program (lowering.model_compiler_tables)
  int N; unsigned long long seed; long long env_offset;
  unsigned char* dirty;                      // [N] set by the step kernels at a device-side restart; cleared here
  const unsigned char* mask;                 // [N] or null: compile these regardless of `dirty` (host-side reset); with all = 1: everyone
  int all;
  unsigned* gen;                             // [N] models this environment has had (the draw counter)
  float* vrec; float* vgt; float* vgpt; int gpt_floats;      // the environment's own tables (slot e)
  float* slack;                              // [12][N] what the pair pass knew: void with another model
  double* draws;                             // [N][n_draw] the values drawn (lm_get_model_draws)
};

void launch(const Args& a, hipStream_t stream);
void replicate(float* dst, const float* src, long long n_per, int N, hipStream_t stream);      // dst[e][i] = src[i]
void iota(int* dst, int N, hipStream_t stream);

}  // namespace lmc
