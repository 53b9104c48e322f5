/* A tiny SIGPROF sampling profiler for the GPU-less harness (tools/host_profile_cpu.py): no perf / valgrind in this image.
 * prof_start(hz) arms ITIMER_REAL (hrtimer resolution; ITIMER_PROF is bound to the 250 Hz tick); every tick stores up to DEPTH return addresses of the interrupted thread; prof_stop() disarms;
 * prof_get(buf, cap) copies the samples out (DEPTH pointers each, 0-terminated).  Symbolisation happens in Python (dladdr). */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <execinfo.h>
#include <signal.h>
#include <string.h>
#include <sys/time.h>
#define DEPTH 24
#define CAP 400000
static void *g_s[CAP][DEPTH];
static volatile int g_n, g_paused;
void prof_pause(int on) { g_paused = on; } /* ticks are dropped while paused */
static void on_prof(int sig) {
    (void) sig;
    if (g_paused) return;
    int i = __sync_fetch_and_add(&g_n, 1);
    if (i >= CAP) return;
    void *tmp[DEPTH + 2];
    int n = backtrace(tmp, DEPTH + 2);
    int k = 0;
    for (int j = 2; j < n && k < DEPTH; j++) g_s[i][k++] = tmp[j]; /* skip the handler + the signal trampoline */
    if (k < DEPTH) g_s[i][k] = 0;
}
void prof_start(int hz) {
    void *w[4];
    backtrace(w, 4); /* load libgcc's unwinder outside the handler */
    g_n = 0;
    g_paused = 0;
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_handler = on_prof;
    sa.sa_flags = SA_RESTART;
    sigaction(SIGALRM, &sa, 0);
    struct itimerval it = {{0, 1000000 / hz}, {0, 1000000 / hz}};
    setitimer(ITIMER_REAL, &it, 0);
}
void prof_stop(void) {
    struct itimerval it = {{0, 0}, {0, 0}};
    setitimer(ITIMER_REAL, &it, 0);
}
int prof_count(void) { return g_n < CAP ? g_n : CAP; }
int prof_depth(void) { return DEPTH; }
void prof_get(void **out) { memcpy(out, g_s, sizeof(void *) * DEPTH * (size_t) prof_count()); }
/* dladdr for Python: symbol name + module of an address (returns 0 when unknown) */
int prof_sym(void *addr, char *name, int ncap, char *mod, int mcap, unsigned long *off) {
    Dl_info di;
    if (!dladdr(addr, &di)) return 0;
    strncpy(name, di.dli_sname ? di.dli_sname : "?", ncap - 1);
    name[ncap - 1] = 0;
    strncpy(mod, di.dli_fname ? di.dli_fname : "?", mcap - 1);
    mod[mcap - 1] = 0;
    *off = (unsigned long) ((char *) addr - (char *) di.dli_fbase);   /* offset in the module: what addr2line wants for a shared object */
    return 1;
}
