// sff_gpu.h — the one process-wide libsffgpu context behind the drop-in header set (include/sff/): the solver
// front ends (forest.h / rrt.h), the FLANN surface (flann/flann.hpp) and the RAPID surface (RAPID.H) all talk to
// the same GPU context, like the reference's process-wide FLANN / RAPID state (Node::globId, Obstacle::rapidId:
// src/primitives.h:493, src/environment.h:90).  Errors follow the reference's convention: message on stdout, exit(1).
#pragma once
#include <cstdlib>
#include <iostream>

#include "../sffgpu.h"

namespace sff_compat {
inline sffgpu_ctx*& gpu_slot() { static sffgpu_ctx* c = nullptr; return c; }
inline sffgpu_ctx* gpu() {
  sffgpu_ctx*& c = gpu_slot();
  if (!c) {
    const char* dev = std::getenv("SFFGPU_DEVICE");
    if (sffgpu_create(dev ? std::atoi(dev) : 0, &c) != SFFGPU_OK) {
      std::cout << "libsffgpu: " << sffgpu_last_error(nullptr) << "\n";
      std::exit(1);
    }
  }
  return c;
}
inline void check(int rc, const char* what) {
  if (rc < 0) {
    std::cout << "libsffgpu: " << what << ": " << sffgpu_last_error(gpu()) << "\n";
    std::exit(1);
  }
}
}  // namespace sff_compat
