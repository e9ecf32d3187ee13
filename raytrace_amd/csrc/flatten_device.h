// flatten_device.h -- device-side construction of the re-flattened BVH records (flatten_device.hip).
#pragma once

#include <hip/hip_runtime.h>

#include "host_scene.h"

namespace rt {

struct DeviceFlat {
  GpuNode2 *nodes2 = nullptr;
  GpuChildQ *nodesq2 = nullptr;
  GpuNode4 *nodes4 = nullptr;    // allocated for n2 records, n4 used
  GpuChildQ *nodesq4 = nullptr;
  GpuTri *tris = nullptr;
  size_t n2 = 0, n4 = 0, ntris = 0;
  int depth4 = 0;
};

// Builds every record array on the current device from the caller's arrays already resident there and the host's
// numbering pass.  On success the caller owns the five device arrays (hipFree).
hipError_t FlattenOnDevice(FlatIndex const &ix, const rt_bvh_node *d_nodes, size_t nnodes, const rt_vec3 *d_verts, const rt_face *d_faces,
                           DeviceFlat *out);

}  // namespace rt
