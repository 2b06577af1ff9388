/* dev aid: LD_PRELOAD this to get the native stack of whatever thread calls abort(). */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <string.h>
#include <unistd.h>
#include <fcntl.h>
static int out_fd(void) { static int fd = -1; if (fd < 0) fd = open("/tmp/abrt_bt.log", O_WRONLY | O_CREAT | O_APPEND, 0644); return fd < 0 ? 2 : fd; }
static void on_abrt(int sig) {
  void* bt[64];
  const char msg[] = "\n=== SIGABRT native backtrace ===\n";
  if (write(out_fd(), msg, sizeof msg - 1) < 0) {}
  int n = backtrace(bt, 64);
  backtrace_symbols_fd(bt, n, out_fd());
  signal(sig, SIG_DFL);
  raise(sig);
}
__attribute__((constructor)) static void install(void) {
  struct sigaction sa;
  memset(&sa, 0, sizeof sa);
  sa.sa_handler = on_abrt;
  sigaction(SIGABRT, &sa, 0);
}
/* libraries that reach abort() through the PLT land here first */
static void dump(const char* why) {
  void* bt[64];
  if (write(out_fd(), why, strlen(why)) < 0) {}
  int n = backtrace(bt, 64);
  backtrace_symbols_fd(bt, n, out_fd());
}
void abort(void) {
  dump("\n=== abort() called ===\n");
  signal(SIGABRT, SIG_DFL);
  raise(SIGABRT);
  _exit(134);
}
void __assert_fail(const char* a, const char* f, unsigned l, const char* fn) {
  dump("\n=== assert failed ===\n");
  if (write(out_fd(), a, strlen(a)) < 0) {}
  if (write(out_fd(), "\n", 1) < 0) {}
  if (write(out_fd(), f, strlen(f)) < 0) {}
  (void)l; (void)fn;
  signal(SIGABRT, SIG_DFL);
  raise(SIGABRT);
  _exit(134);
}
