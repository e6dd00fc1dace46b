// C ABI of the MI355X-native encode_batch path (include/tokenizers_amd.h): tokenizer handle,
// HBM workspace, stream-ordered kernel pipeline.  No tokenisation logic lives here -- it only
// sequences the kernels of kernels.hip.
#include "../../include/tokenizers_amd.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <stdexcept>
#include <thread>
#include <chrono>
#include <dlfcn.h>
#include <pthread.h>
#include <string>
#include <vector>

#include "host_model.hpp"
#include "kernels.hpp"
#include "overflow_core.hpp"
#include "bert_norm_core.hpp"

using namespace tkamd;
static_assert(TEXT_PAD == TKAMD_TEXT_PAD, "the kernels rely on the slack the ABI promises");

// The parts of this translation unit, in order (each is #included here and not compiled on its own -- like kernels.hip and its kernels/*.hip):
#include "capi/support.cpp"      // errors, the fork() guard, RCCL opened at first use, the rendezvous objects of a sharded call, DevBuf
#include "capi/handle.cpp"      // Workspace, tkamd_tokenizer, the pinned-block pool, tkamd_batch / tkamd_text
#include "capi/tables.cpp"      // stage timers, the device tables made at load: upload, the load-time proof of the whole-word table, the short-word and hot tables
#include "capi/pipeline.cpp"      // the kernel sequence of one batch (run_pipeline), its queues and workspace sizes, the synchronisation and error mapping
#include "capi/pool.cpp"      // the workspace pool of a handle
#include "capi/api_handle.cpp"      // extern "C": handles, device-buffer entries
#include "capi/sharding.cpp"      // one host-entry call over the devices of a multi-device handle
#include "capi/host_entry.cpp"      // the host-buffer entries (sliced, paced, mixed, words) and the result accessors
#include "capi/decode.cpp"      // decode_batch
#include "capi/probes.cpp"      // host-side probes of the load-time tables, switches, measurement hooks
