// The library's side streams: created ONCE per process and device, shared by every object that forks work off the caller's stream (the SPLIT-VAE plan's
// weight-gradient streams, the SPLIT-SPAIR tape's lanes, the SPLIT-GMVAE step's second encoder stream), never destroyed.
//
// HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues in creation order.  With a stream per object, the second model of a process (bench.py's rows, a
// training loop that also evaluates at another batch size) created its side stream as the process's fourth or fifth one -- on the hardware queue of the
// compute stream it is supposed to run beside -- and every cross-stream event between the two cost a round trip through the host: 1.19 -> 3.84 ms for the
// SPLIT-GMVAE step, 2.65 -> 7.0 ms for SPLIT-SPAIR (profiles/r06_gm_streams.txt).  Two or three shared streams keep the mapping fixed: caller's stream + 2 =
// the three hardware queues split_vae_amd.configure_hw_queues() asks for.  Objects used from several host threads share the streams too: every consumer
// orders its work with its own fork / join events, so sharing costs overlap, never correctness.
#include <mutex>
#include <map>
#include <stdlib.h>
#include "common.hip.h"
#include "kernels.h"

namespace {
struct DevStreams { hipStream_t s[SV_SHARED_STREAMS] = {}; };
std::mutex g_mu;
std::map<int, DevStreams> g_streams;
}  // namespace

// side stream k (0 .. SV_SHARED_STREAMS - 1) of the current device, or nullptr when it cannot be created
hipStream_t sv_shared_stream(int k) {
  if (k < 0 || k >= SV_SHARED_STREAMS) return nullptr;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  std::lock_guard<std::mutex> lk(g_mu);
  DevStreams& d = g_streams[dev];
  for (int i = 0; i <= k; ++i) {                 // in index order: stream i is always the (i + 1)-th stream the library creates on the device
    if (d.s[i]) continue;
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    static const bool normal = getenv("SV_SIDE_PRIO_NORMAL") != nullptr;     // (A/B: the side streams at the default priority)
    if (hipStreamCreateWithPriority(&d.s[i], hipStreamNonBlocking, normal ? 0 : lo) != hipSuccess) { d.s[i] = nullptr; return nullptr; }
  }
  return d.s[k];
}

extern "C" int sv_side_stream(int32_t index, void** stream) {
  if (!stream) return SV_E_BADARG;
  hipStream_t s = sv_shared_stream(index);
  if (!s) return index < 0 || index >= SV_SHARED_STREAMS ? SV_E_BADARG : (int)hipGetLastError();
  *stream = (void*)s;
  return SV_OK;
}
