// vil_internal.h -- entry points shared between the translation units of libvilsolve.so that are NOT part of the public
// C-ABI (include/*.h): they take device pointers.
#pragma once
#include "../../include/vilsolve.h"

// LiDAR point factors that already live on the device, structure-of-arrays, all attached to window pose 0:
//   plane_soa[q * plane_stride + f], q < 7 (cp n d) ; edge_soa[q * edge_stride + f], q < 9 (cp a b)
// (the layout the sweep reads; strides are multiples of 32).  Used by vmap_align: the association kernels write these
// tables and the solver consumes them without a host round trip.
struct vil_device_lidar { const double* plane_soa; int plane_stride; const double* edge_soa; int edge_stride; };

// vil_solve with p->n_plane / p->n_edge factors taken from `dl` (p->plane_* / p->edge_* are ignored).  The tables must stay
// valid until the call returns; they are written by work enqueued on `producer` (the call waits for it on the device).
extern "C" int vil_solve_device_lidar(vil_ctx* ctx, const vil_problem* p, const vil_device_lidar* dl, void* producer_stream,
                                      vil_state* s, const vil_options* o, vil_summary* sum);

// vilpreint.hip: pre-integrates ONE interval whose samples (dt[ns], acc[3 ns], gyr[3 ns]), header (acc0 gyr0 lin_ba lin_bg) and
// 287-double record all live on the device -- the IMU slots of the resident window (vil_window.hpp).  Asynchronous on `stream`.
void vpre_launch_slot(hipStream_t stream, int ns, const double* dt, const double* acc, const double* gyr, const double* hdr12, const double* noise4, double* rec);
