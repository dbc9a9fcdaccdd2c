/* A poor man's perf: SIGPROF at 10 kHz, program counters into a buffer, dumped as "module+offset" lines.
 * gcc -O2 -shared -fPIC -o pcsample.so pcsample.c ; ctypes: pcsample_start(), pcsample_stop(path).  (No perf in this image.) */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <signal.h>
#include <stdio.h>
#include <string.h>
#include <sys/time.h>
#include <time.h>
#include <ucontext.h>

#define CAP (1 << 20)
static void* pcs[CAP];
static volatile int n_pcs;
static timer_t tm;

static void on_prof(int sig, siginfo_t* si, void* uc_) {
  (void)sig; (void)si;
  ucontext_t* uc = (ucontext_t*)uc_;
  int i = __sync_fetch_and_add(&n_pcs, 1);
  if (i < CAP) pcs[i] = (void*)uc->uc_mcontext.gregs[REG_RIP];
}

void pcsample_start(void) {
  struct sigaction sa;
  memset(&sa, 0, sizeof(sa));
  sa.sa_sigaction = on_prof;
  sa.sa_flags = SA_SIGINFO | SA_RESTART;
  sigaction(SIGPROF, &sa, 0);
  n_pcs = 0;
  /* ITIMER_PROF ticks at the kernel's HZ (250/s here): a POSIX timer on the monotonic clock at 10 kHz instead */
  struct sigevent ev;
  memset(&ev, 0, sizeof(ev));
  ev.sigev_notify = SIGEV_SIGNAL;
  ev.sigev_signo = SIGPROF;
  timer_create(CLOCK_MONOTONIC, &ev, &tm);
  struct itimerspec its = {{0, 100000}, {0, 100000}};
  timer_settime(tm, 0, &its, 0);
}

void pcsample_stop(const char* path) {
  timer_delete(tm);
  signal(SIGPROF, SIG_IGN);
  FILE* f = fopen(path, "w");
  int n = n_pcs < CAP ? n_pcs : CAP;
  for (int i = 0; i < n; ++i) {
    Dl_info di;
    if (dladdr(pcs[i], &di) && di.dli_fname)
      fprintf(f, "%s %lx %s\n", di.dli_fname, (unsigned long)((char*)pcs[i] - (char*)di.dli_fbase), di.dli_sname ? di.dli_sname : "?");
    else
      fprintf(f, "? %p ?\n", pcs[i]);
  }
  fclose(f);
}
