// rccl_dyn.hpp -- RCCL bound at run time.
//
// The engine needs exactly one collective: the sum all-reduce of the per-GPU
// charge vector (nx doubles) that replaces MPI_Allreduce at
// src/pic1dp_interaction.F90:132.  RCCL is resolved with dlopen so that
//  * a single-GPU process never loads it, and
//  * a process that already has an RCCL mapped (e.g. a Python host that
//    imported torch, which ships librccl.so.1) shares that copy instead of
//    mapping a second one with clashing symbols.
#pragma once
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <string>

namespace pic1dp {

struct RcclApi {
  void *handle = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;

  bool load(std::string &err) {
    if (handle) return true;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *n : names) {
      handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);  // already mapped?
      if (handle) break;
    }
    for (int i = 0; !handle && i < 3; ++i) handle = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
    if (!handle) {
      err = std::string("cannot load librccl: ") + dlerror();
      return false;
    }
    GetUniqueId = reinterpret_cast<decltype(GetUniqueId)>(dlsym(handle, "ncclGetUniqueId"));
    CommInitRank = reinterpret_cast<decltype(CommInitRank)>(dlsym(handle, "ncclCommInitRank"));
    CommDestroy = reinterpret_cast<decltype(CommDestroy)>(dlsym(handle, "ncclCommDestroy"));
    AllReduce = reinterpret_cast<decltype(AllReduce)>(dlsym(handle, "ncclAllReduce"));
    GetErrorString = reinterpret_cast<decltype(GetErrorString)>(dlsym(handle, "ncclGetErrorString"));
    if (!GetUniqueId || !CommInitRank || !CommDestroy || !AllReduce || !GetErrorString) {
      err = "librccl lacks a required symbol";
      handle = nullptr;  // not usable: the next load() reports the failure again instead of returning true
      GetUniqueId = nullptr, CommInitRank = nullptr, CommDestroy = nullptr, AllReduce = nullptr, GetErrorString = nullptr;
      return false;
    }
    return true;
  }
};

inline RcclApi &rccl() {
  static RcclApi api;
  return api;
}

}  // namespace pic1dp
