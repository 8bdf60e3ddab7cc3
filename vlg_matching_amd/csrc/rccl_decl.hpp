// The few declarations of <rccl/rccl.h> that comm.hip calls through dlsym'd pointers (libvlg_hip.so does not link RCCL: the
// process's own copy is bound at run time).  They live in their own namespace so that rccl_abi_check.cpp can include the real
// header beside them and static_assert that sizes, enum values and signatures still agree -- a silent drift would corrupt
// communicators instead of failing the build.
#pragma once
#include <cstddef>
#include <hip/hip_runtime.h>

namespace vlg_rccl {

struct Comm;
typedef Comm* comm_t;                                         // ncclComm_t (an opaque pointer)
constexpr size_t kUniqueIdBytes = 128;                        // NCCL_UNIQUE_ID_BYTES
struct UniqueId { char internal[kUniqueIdBytes]; };           // ncclUniqueId (passed BY VALUE to ncclCommInitRank)
enum : int { kSuccess = 0 };                                  // ncclSuccess
enum : int { kInt8 = 0, kUint8 = 1, kInt32 = 2, kUint32 = 3, kInt64 = 4, kUint64 = 5 };      // ncclDataType_t
enum : int { kSum = 0 };                                      // ncclRedOp_t

// result and enum parameters travel as int (the C enums of rccl.h have int as their underlying type: checked)
typedef int (*GetUniqueId_fn)(UniqueId*);
typedef int (*CommInitRank_fn)(comm_t*, int, UniqueId, int);
typedef int (*CommDestroy_fn)(comm_t);
typedef int (*CommCount_fn)(comm_t, int*);
typedef int (*CommUserRank_fn)(comm_t, int*);
typedef int (*Broadcast_fn)(const void*, void*, size_t, int, int, comm_t, hipStream_t);
typedef int (*AllReduce_fn)(const void*, void*, size_t, int, int, comm_t, hipStream_t);
typedef int (*AllGather_fn)(const void*, void*, size_t, int, comm_t, hipStream_t);
typedef int (*Send_fn)(const void*, size_t, int, int, comm_t, hipStream_t);
typedef int (*Recv_fn)(void*, size_t, int, int, comm_t, hipStream_t);
typedef int (*Group_fn)();
typedef const char* (*GetErrorString_fn)(int);

}  // namespace vlg_rccl
