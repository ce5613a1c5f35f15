// f360_state.h -- the seam between the two device translation units of the library: rgbd360_api.hip (contexts, alignment) owns a
// context; rgbd360_frame360.hip (Frame360 stages) keeps its scratch in an F360State the context points to and reaches the context only
// through the two functions below.  Neither unit sees the other's structs or kernels.
#pragma once
#include <hip/hip_runtime.h>

struct rgbd360_ctx;
struct F360State;

// rgbd360_frame360.hip
F360State* f360_state_create(int device, hipStream_t stream);      // nullptr: out of memory
void f360_state_destroy(F360State* s);
// rgbd360_api.hip
F360State* rgbd360_ctx_f360(rgbd360_ctx* ctx);                      // the context's state, created on first use (on its device and stream)
void rgbd360_ctx_set_error(rgbd360_ctx* ctx, const char* msg);      // what rgbd360_last_error(ctx) returns next
