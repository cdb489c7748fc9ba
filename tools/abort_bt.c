/* Debug aid: LD_PRELOAD this to get a C-level backtrace on SIGABRT (glibc heap checks, std::terminate) even after the
 * interpreter's own fault handler is gone (static destructors at exit).  gcc -shared -fPIC -o abort_bt.so abort_bt.c */
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>
static void on_abort(int s) {
  void* f[96];
  int n = backtrace(f, 96);
  static const char m[] = "[abort_bt] SIGABRT backtrace:\n";
  (void)!write(2, m, sizeof(m) - 1);
  backtrace_symbols_fd(f, n, 2);
  signal(s, SIG_DFL);
  raise(s);
}
__attribute__((constructor)) static void abort_bt_init(void) { signal(SIGABRT, on_abort); }
