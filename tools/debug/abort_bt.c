// LD_PRELOAD helper: native backtrace when somebody calls abort() / fails an assert (who aborts the process?).
// gcc -shared -fPIC -o abort_bt.so abort_bt.c ; LD_PRELOAD=./abort_bt.so python -m pytest -p no:faulthandler ...
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <unistd.h>
static void dump(const char *why) {
  void *bt[64];
  int n = backtrace(bt, 64);
  dprintf(2, "\n=== abort_bt: %s, native backtrace ===\n", why);
  backtrace_symbols_fd(bt, n, 2);
  dprintf(2, "=== end ===\n");
}
void abort(void) {
  dump("abort() called");
  signal(SIGABRT, SIG_DFL);
  raise(SIGABRT);
  _exit(134);
}
void __assert_fail(const char *expr, const char *file, unsigned line, const char *fn) {
  dprintf(2, "assertion failed: %s (%s:%u %s)\n", expr, file, line, fn);
  abort();
}
static void handler(int sig) { dump("signal"); signal(sig, SIG_DFL); raise(sig); }
__attribute__((constructor)) static void init(void) { signal(SIGSEGV, handler); signal(SIGBUS, handler); }
