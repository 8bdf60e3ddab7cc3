// Compile-only: the RCCL declarations comm.hip binds by hand (rccl_decl.hpp) against the real <rccl/rccl.h> of this ROCm.
// Nothing here is linked into libvlg_hip.so; `make` compiles it to an object that is thrown away.  A mismatch fails the build.
#include <type_traits>
#include <rccl/rccl.h>
#include "rccl_decl.hpp"

using namespace vlg_rccl;

static_assert(sizeof(UniqueId) == sizeof(ncclUniqueId) && alignof(UniqueId) == alignof(ncclUniqueId), "ncclUniqueId");
static_assert(kUniqueIdBytes == NCCL_UNIQUE_ID_BYTES, "NCCL_UNIQUE_ID_BYTES");
static_assert(std::is_trivially_copyable<ncclUniqueId>::value, "ncclUniqueId is passed by value as plain bytes");
static_assert(sizeof(comm_t) == sizeof(ncclComm_t), "ncclComm_t is a pointer");
static_assert((int)ncclSuccess == kSuccess, "ncclSuccess");
static_assert((int)ncclInt8 == kInt8 && (int)ncclUint8 == kUint8 && (int)ncclInt32 == kInt32 && (int)ncclUint32 == kUint32 &&
              (int)ncclInt64 == kInt64 && (int)ncclUint64 == kUint64, "ncclDataType_t values");
static_assert((int)ncclSum == kSum, "ncclRedOp_t values");
static_assert(sizeof(ncclResult_t) == sizeof(int) && sizeof(ncclDataType_t) == sizeof(int) && sizeof(ncclRedOp_t) == sizeof(int),
              "the enums travel as int");

// signatures: same parameter lists up to (enum <-> int) and (ncclComm_t <-> comm_t), which share size and calling convention
template <class Mine, class Theirs> struct same_shape : std::false_type {};
template <class R1, class... A1, class R2, class... A2>
struct same_shape<R1 (*)(A1...), R2 (*)(A2...)>
    : std::integral_constant<bool, sizeof...(A1) == sizeof...(A2) && sizeof(R1) == sizeof(R2)> {};
template <class A, class B> constexpr bool arg_ok() { return sizeof(A) == sizeof(B) && std::is_pointer<A>::value == std::is_pointer<B>::value; }
template <class Mine, class Theirs> struct args_ok;
template <class R1, class... A1, class R2, class... A2>
struct args_ok<R1 (*)(A1...), R2 (*)(A2...)> {
    static constexpr bool all()
    {
        bool ok = true;
        const bool each[] = {true, arg_ok<A1, A2>()...};
        for (bool b : each) ok = ok && b;
        return ok;
    }
};
#define VLG_CHECK_FN(mine, theirs)                                                                           \
    static_assert(same_shape<mine, decltype(&theirs)>::value, #theirs ": parameter count / result size");   \
    static_assert(args_ok<mine, decltype(&theirs)>::all(), #theirs ": parameter sizes")
VLG_CHECK_FN(GetUniqueId_fn, ncclGetUniqueId);
VLG_CHECK_FN(CommInitRank_fn, ncclCommInitRank);
VLG_CHECK_FN(CommDestroy_fn, ncclCommDestroy);
VLG_CHECK_FN(CommCount_fn, ncclCommCount);
VLG_CHECK_FN(CommUserRank_fn, ncclCommUserRank);
VLG_CHECK_FN(Broadcast_fn, ncclBroadcast);
VLG_CHECK_FN(AllReduce_fn, ncclAllReduce);
VLG_CHECK_FN(AllGather_fn, ncclAllGather);
VLG_CHECK_FN(Send_fn, ncclSend);
VLG_CHECK_FN(Recv_fn, ncclRecv);
VLG_CHECK_FN(Group_fn, ncclGroupStart);
VLG_CHECK_FN(Group_fn, ncclGroupEnd);
VLG_CHECK_FN(GetErrorString_fn, ncclGetErrorString);
