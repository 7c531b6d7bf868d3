/* LD_PRELOAD aid for crash hunts on the GPU box: a SIGSEGV / SIGBUS / SIGABRT in ANY thread prints the thread's
 * backtrace (glibc backtrace_symbols_fd: async-signal-safe enough for a dying process) and the fault address,
 * then re-raises with the default action so that the exit status is unchanged.
 *   gcc -O1 -g -shared -fPIC -o /tmp/segv_trap.so tools/probes/segv_trap.c -ldl */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <string.h>
#include <unistd.h>
#include <sys/syscall.h>

static void on_fault(int sig, siginfo_t* si, void* uc) {
    (void)uc;
    char line[160];
    int n = snprintf(line, sizeof line, "\n[segv_trap] signal %d addr %p pid %d tid %ld\n", sig, si ? si->si_addr : 0,
                     (int)getpid(), (long)syscall(SYS_gettid));
    if (write(2, line, (size_t)n) < 0) {}
    void* bt[64];
    int k = backtrace(bt, 64);
    backtrace_symbols_fd(bt, k, 2);
    /* the mappings tell which library an unnamed frame belongs to */
    FILE* f = fopen("/proc/self/maps", "r");
    if (f) {
        char l[512];
        while (fgets(l, sizeof l, f))
            if (strstr(l, "r-xp") || strstr(l, "r-x")) fputs(l, stderr);
        fclose(f);
    }
    signal(sig, SIG_DFL);
    raise(sig);
}

__attribute__((constructor)) static void install(void) {
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_sigaction = on_fault;
    sa.sa_flags = SA_SIGINFO | SA_NODEFER;
    static char stack[1 << 16];
    stack_t ss;
    ss.ss_sp = stack;
    ss.ss_size = sizeof stack;
    ss.ss_flags = 0;
    sigaltstack(&ss, 0);
    sa.sa_flags |= SA_ONSTACK;
    sigaction(SIGSEGV, &sa, 0);
    sigaction(SIGBUS, &sa, 0);
    sigaction(SIGABRT, &sa, 0);
}
