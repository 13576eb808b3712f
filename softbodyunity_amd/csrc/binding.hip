// binding.hip — error state, the run-time binding of RCCL and what the plugin reports about the ROCm stack it runs on
//
// No reference counterpart exists (/root/reference/README.md:1 is the whole reference tree); the exported functions are the
// [BUILDER-DEFINED] boundary of SURVEY.md §8b (include/softbody*.h).
#include "solver_internal.hpp"

#include <csignal>
#include <unistd.h>

namespace sbi {

namespace { thread_local std::string g_err; }
int fail(int code, const std::string &msg) { g_err = msg; return code; }
const char *last_error_text() { return g_err.c_str(); }

// RCCL is NOT a link-time dependency. A world == 1 host never loads it (the library is 570 MB); a world > 1 host binds, on
// first use, the librccl.so.1 that is ALREADY in the process when there is one -- a host that imported PyTorch first brought
// PyTorch's own RCCL together with PyTorch's own HIP runtime, which this plugin's libamdhip64.so.7 dependency resolved to as
// well, and a second ROCm stack in one process is the one thing that must not happen -- and the system's otherwise (the
// plugin's RUNPATH: /opt/rocm/lib). What was bound is reported by sb_runtime_info and decides which schedules are admitted.
RcclApi &rccl(bool required) {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *names[] = {"librccl.so.1", "librccl.so"};
        for (const char *nm : names) if (!api.handle) { api.handle = dlopen(nm, RTLD_NOW | RTLD_NOLOAD); api.was_resident = api.handle != nullptr; }
        for (const char *nm : names) if (!api.handle) api.handle = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
        if (!api.handle) { const char *e = dlerror(); api.error = std::string("RCCL could not be loaded: ") + (e ? e : "librccl.so.1 not found"); return; }
        bool all = true;
        auto sym = [&](auto &fn, const char *name) { fn = reinterpret_cast<std::remove_reference_t<decltype(fn)>>(dlsym(api.handle, name)); all &= fn != nullptr; };
        sym(api.GetVersion, "ncclGetVersion"); sym(api.GetUniqueId, "ncclGetUniqueId"); sym(api.CommInitRank, "ncclCommInitRank");
        sym(api.CommDestroy, "ncclCommDestroy"); sym(api.GetErrorString, "ncclGetErrorString"); sym(api.GroupStart, "ncclGroupStart");
        sym(api.GroupEnd, "ncclGroupEnd"); sym(api.Send, "ncclSend"); sym(api.Recv, "ncclRecv"); sym(api.AllGather, "ncclAllGather");
        if (!all) { api.error = "RCCL library lacks a symbol the plugin needs"; api.handle = nullptr; return; }
        Dl_info di;
        if (dladdr(reinterpret_cast<void *>(api.Send), &di) && di.dli_fname) api.path = di.dli_fname;
        if (api.GetVersion(&api.version) != ncclSuccess) api.version = 0;
        // the plugin uses only calls whose signatures have not changed since NCCL 2.7 (send/recv); refuse anything older
        if (api.version < 20700) { api.error = "RCCL " + std::to_string(api.version) + " is older than 2.7 (no ncclSend/ncclRecv)"; api.handle = nullptr; }
    });
    if (required && !api.ok()) throw HipError(SB_ERR_UNSUPPORTED, api.error);
    return api;
}

// HIP runtimes older than 7.2 recurse without bound in hipStreamEndCapture when a captured stream that was itself forked
// from the origin (the exchange stream of the overlapped schedule) is forked again (RCCL's internal stream joins the capture
// from the stream it is called on): the list of parallel capture streams becomes cyclic. Found with a native backtrace on
// PyTorch's bundled HIP 7.0.51831 (profiles/r03a_overlap_capture_backtrace.txt); the same process on HIP 7.2.26015 is fine.
int hip_runtime_version() {
    static const int v = [] { int x = 0; return hipRuntimeGetVersion(&x) == hipSuccess ? x : 0; }();
    return v;
}
bool capture_overlap_ok() { return hip_runtime_version() >= 70200000; }

int set_device(const sb_solver *s) {
    hipError_t e = hipSetDevice(s->desc.device);
    if (e != hipSuccess) return fail(SB_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(e));
    return SB_OK;
}

}  // namespace sbi

using namespace sbi;

extern "C" {

const char *sb_last_error(void) { return last_error_text(); }
int sb_abi_version(void) { return SB_ABI_VERSION; }

int sb_runtime_info(sb_runtime_info_t *out) {
    if (!out) return fail(SB_ERR_INVALID_ARG, "sb_runtime_info: null argument");
    std::memset(out, 0, sizeof(*out));
    return guarded([&]() -> int {
        out->hip_runtime_version = hip_runtime_version();
        int drv = 0;
        if (hipDriverGetVersion(&drv) == hipSuccess) out->hip_driver_version = drv;
        Dl_info di;
        if (dladdr(reinterpret_cast<void *>(&hipRuntimeGetVersion), &di) && di.dli_fname) std::snprintf(out->hip_library, sizeof out->hip_library, "%s", di.dli_fname);
        out->rccl_header_version = NCCL_VERSION_CODE;
        RcclApi &R = rccl(false);
        if (R.ok()) {
            out->rccl_version = R.version;
            out->rccl_was_resident = R.was_resident ? 1 : 0;
            std::snprintf(out->rccl_library, sizeof out->rccl_library, "%s", R.path.c_str());
            out->capture_serial_ok = R.version >= 22606 ? 1 : 0;
            out->capture_overlap_ok = (R.version >= 22606 && capture_overlap_ok()) ? 1 : 0;
        }
        return SB_OK;
    });
}

// ---- last words (softbody_debug.h): ONE result line survives a fatal signal ----------------------------------------------------------
namespace {
constexpr int kLastWordSignals[] = {SIGSEGV, SIGBUS, SIGABRT, SIGFPE, SIGILL, SIGTERM};
constexpr int kNumLastWordSignals = (int)(sizeof(kLastWordSignals) / sizeof(kLastWordSignals[0]));
// two buffers used alternately, so that the handler never sees a half-written text: `g_lw_text` is switched last
char *g_lw_buf[2] = {nullptr, nullptr};
size_t g_lw_cap[2] = {0, 0};
std::atomic<const char *> g_lw_text{nullptr};
std::atomic<size_t> g_lw_len{0};
std::atomic<int> g_lw_fd{-1}, g_lw_exit{70};
int g_lw_next = 0;
bool g_lw_installed = false;
struct sigaction g_lw_old[kNumLastWordSignals];
void last_words_handler(int) {
    const char *t = g_lw_text.load(std::memory_order_acquire);
    size_t n = g_lw_len.load(std::memory_order_acquire);
    const int fd = g_lw_fd.load(std::memory_order_acquire);
    while (t && n > 0 && fd >= 0) {
        const ssize_t w = ::write(fd, t, n);
        if (w <= 0) break;
        t += w; n -= (size_t)w;
    }
    _exit(g_lw_exit.load(std::memory_order_acquire));
}
}  // namespace

int sb_debug_last_words(int32_t fd, const char *text, int64_t len, int32_t exit_code) {
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    if (!text || len <= 0) {       // remove the registration
        g_lw_text.store(nullptr, std::memory_order_release); g_lw_len.store(0, std::memory_order_release);
        if (g_lw_installed) {
            for (int k = 0; k < kNumLastWordSignals; ++k) (void)sigaction(kLastWordSignals[k], &g_lw_old[k], nullptr);
            g_lw_installed = false;
        }
        return SB_OK;
    }
    if (fd < 0) return fail(SB_ERR_INVALID_ARG, "sb_debug_last_words: bad file descriptor");
    const int q = g_lw_next; g_lw_next ^= 1;
    if (g_lw_cap[q] < (size_t)len) {
        char *nb = static_cast<char *>(std::realloc(g_lw_buf[q], (size_t)len));
        if (!nb) return fail(SB_ERR_NOMEM, "sb_debug_last_words: out of host memory");
        g_lw_buf[q] = nb; g_lw_cap[q] = (size_t)len;
    }
    std::memcpy(g_lw_buf[q], text, (size_t)len);
    g_lw_fd.store(fd, std::memory_order_release); g_lw_exit.store(exit_code, std::memory_order_release);
    g_lw_len.store(0, std::memory_order_release);                    // (a handler between the two stores writes nothing rather than a torn text)
    g_lw_text.store(g_lw_buf[q], std::memory_order_release);
    g_lw_len.store((size_t)len, std::memory_order_release);
    if (!g_lw_installed) {
        struct sigaction sa;
        std::memset(&sa, 0, sizeof sa);
        sa.sa_handler = last_words_handler;
        sigemptyset(&sa.sa_mask);
        for (int k = 0; k < kNumLastWordSignals; ++k) (void)sigaction(kLastWordSignals[k], &sa, &g_lw_old[k]);
        g_lw_installed = true;
    }
    return SB_OK;
}

}  // extern "C"
