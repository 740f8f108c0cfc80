// Launch programs: the library's own record / replay of a launch sequence over explicit HIP streams.
//
// Every kernel launch of the library goes through ASE_LAUNCH.  While a thread records a program
// (ase_hip_prog_begin .. ase_hip_prog_end) nothing is launched: each call site stores a closure {kernel, grid, block,
// arguments by value} together with the stream it addressed, and fork / join points (ase_hip_mark / ase_hip_wait) store
// event records / waits.  ase_hip_prog_launch replays the list on the SAME streams with 4-5 us of host work per entry (measured).
// Unlike a captured hipGraph the mapping of branches to streams (hence to hardware queues) is ours and fixed, and unlike
// eager launches from Python the host never falls behind the GPU.
#pragma once
#include <hip/hip_runtime.h>
#include <functional>

struct AseProgram;
AseProgram* ase_prog_recording();        // this thread's recording target, or nullptr
void ase_prog_push(AseProgram* pg, hipStream_t stream, std::function<void(hipStream_t)>&& fn);

#define ASE_LAUNCH(kern, grid, block, shmem, stream, ...)                                              \
    do {                                                                                               \
        if (AseProgram* pg__ = ase_prog_recording()) {                                                 \
            ase_prog_push(pg__, (hipStream_t)(stream), [=](hipStream_t s__) {                          \
                hipLaunchKernelGGL(kern, grid, block, shmem, s__, __VA_ARGS__);                        \
            });                                                                                        \
        } else {                                                                                       \
            hipLaunchKernelGGL(kern, grid, block, shmem, (hipStream_t)(stream), __VA_ARGS__);          \
        }                                                                                              \
    } while (0)
