// image_matching_amd/csrc/capi_internal.h — handle types and error plumbing shared by the extern "C" translation units
// (capi.cpp: contexts, keys, ciphertexts, roles; group.cpp: the sharded multi-GPU sender).
#pragma once
#include <atomic>
#include <string>

#include "../../include/hydia.h"
#include "client.h"
#include "hydia_core.h"

// A context outlives its ciphertext handles: every hydia_ct pins the context it lives in, and hydia_ctx_destroy on a context
// that still has handles only marks it — the last hydia_ct_free completes the destruction (the HBM pool the handles return
// their memory to is part of the context).
struct hydia_ctx {
    hydia::Context cx;
    // one reference of the context itself (dropped by hydia_ctx_destroy) + one per live hydia_ct: whoever drops the last one deletes
    // the context.  Atomic: a binding's finaliser may free a handle on another thread than the one using the context — which is all
    // such a free does concurrently: it returns the handle's block to the context's pool, whose free lists are guarded by a mutex
    // (Pool::mu_), and hipSetDevice only sets the calling thread's current device.  Everything else on a context (queries, key
    // loading, the database) stays single-threaded: thread-compatible, not thread-safe, like the reference's classes.
    std::atomic<long> refs{1};
    hydia_ctx(const hydia::Params &p, int dev) : cx(p, dev) {}
};
struct hydia_ct {
    hydia::Ct c;
    hydia_ctx *owner = nullptr;
};

int hydia_fail(int code, const std::string &msg);  // records the calling thread's last error and returns `code`

#define API_BEGIN try {
#define API_END                                                                              \
    }                                                                                        \
    catch (const hydia::DeviceError &e) { return hydia_fail(HYDIA_ERR_DEVICE, e.what()); }   \
    catch (const hydia::StateError &e) { return hydia_fail(HYDIA_ERR_STATE, e.what()); }     \
    catch (const std::runtime_error &e) { return hydia_fail(HYDIA_ERR_ARG, e.what()); }      \
    catch (const std::exception &e) { return hydia_fail(HYDIA_ERR_INTERNAL, e.what()); }
#define REQUIRE(cond, msg) \
    if (!(cond)) return hydia_fail(HYDIA_ERR_ARG, msg)

// handles that cross the C-ABI are always compact ([count][poly][limb][N], nl == lstride) and owning
inline hydia_ct *wrap(hydia_ctx *owner, hydia::Ct &&c) {
    hydia_ct *h = new hydia_ct;
    if (c.view || !c.compact()) h->c = c.ctx->clone(c);
    else h->c = std::move(c);
    h->owner = owner;
    owner->refs.fetch_add(1);
    return h;
}
// HIP's current device is per host thread: every entry point that takes a context selects the context's GPU first, so one
// process can drive contexts on several GPUs (the worker threads of a shard group each bind their own)
inline void use_device(const hydia_ctx *ctx) {
    if (ctx) (void)hipSetDevice(ctx->cx.device);
}
hydia::Params hydia_to_params(const hydia_params *p);
