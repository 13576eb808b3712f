/* LD_PRELOAD helper for one-off diagnosis on the GPU box: prints a native backtrace (module + offset, nearest exported
 * symbol) when the process takes SIGSEGV / SIGBUS / SIGABRT, then exits with 139. Build: gcc -O1 -g -shared -fPIC -o
 * tools/probe/libsegv_bt.so tools/probe/segv_bt.c -ldl. Not part of the product. */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

static void handler(int sig, siginfo_t *info, void *ctx) {
    (void)ctx;
    void *frames[96];
    char line[512];
    int n = backtrace(frames, 96);
    int len = snprintf(line, sizeof line, "\n=== segv_bt: signal %d, fault address %p, %d frames ===\n", sig, info ? info->si_addr : NULL, n);
    if (write(2, line, (size_t)len) < 0) _exit(139);
    for (int i = 0; i < n; ++i) {
        Dl_info d;
        memset(&d, 0, sizeof d);
        if (dladdr(frames[i], &d) && d.dli_fname)
            len = snprintf(line, sizeof line, "#%02d %p %s +0x%lx  %s+0x%lx\n", i, frames[i], d.dli_fname,
                           (unsigned long)((char *)frames[i] - (char *)d.dli_fbase), d.dli_sname ? d.dli_sname : "?",
                           d.dli_saddr ? (unsigned long)((char *)frames[i] - (char *)d.dli_saddr) : 0ul);
        else
            len = snprintf(line, sizeof line, "#%02d %p ?\n", i, frames[i]);
        if (write(2, line, (size_t)len) < 0) break;
    }
    _exit(139);
}

__attribute__((constructor)) static void install(void) {
    static char stack[1 << 16];
    stack_t ss;
    ss.ss_sp = stack; ss.ss_size = sizeof stack; ss.ss_flags = 0;
    sigaltstack(&ss, NULL);
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_sigaction = handler;
    sa.sa_flags = SA_SIGINFO | SA_ONSTACK | SA_RESETHAND;
    sigaction(SIGSEGV, &sa, NULL);
    sigaction(SIGBUS, &sa, NULL);
    sigaction(SIGABRT, &sa, NULL);
    void *warm[4];
    backtrace(warm, 4);      /* loads libgcc_s now, not inside the handler */
}
