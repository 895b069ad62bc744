/* TEST INFRASTRUCTURE.  A SIGABRT handler that appends the native stack of the aborting thread (and the maps of the loaded
 * shared objects, so that the addresses can be resolved with addr2line afterwards) to a file, then hands over to the handler
 * that was installed before it (Python's faulthandler) and finally lets the abort take its course.  Loaded by
 * tests/conftest.py; the file lands under gpurun_out/, which gpurun merges back from the GPU box. */
#define _GNU_SOURCE
#include <execinfo.h>
#include <fcntl.h>
#include <signal.h>
#include <string.h>
#include <unistd.h>

static char g_path[1024];
static struct sigaction g_prev;

static void put(int fd, const char *s) { (void)!write(fd, s, strlen(s)); }

static void on_abort(int sig, siginfo_t *info, void *uctx) {
  int fd = open(g_path, O_WRONLY | O_CREAT | O_APPEND, 0644);
  if (fd >= 0) {
    void *frames[96];
    const int n = backtrace(frames, 96);
    put(fd, "==== SIGABRT: native stack of the aborting thread ====\n");
    backtrace_symbols_fd(frames, n, fd);
    put(fd, "==== /proc/self/maps (executable mappings) ====\n");
    int m = open("/proc/self/maps", O_RDONLY);
    if (m >= 0) {
      char buf[4096];
      ssize_t k;
      while ((k = read(m, buf, sizeof buf)) > 0) (void)!write(fd, buf, (size_t)k);
      close(m);
    }
    close(fd);
  }
  if (g_prev.sa_flags & SA_SIGINFO) {
    if (g_prev.sa_sigaction) g_prev.sa_sigaction(sig, info, uctx);
  } else if (g_prev.sa_handler != SIG_DFL && g_prev.sa_handler != SIG_IGN && g_prev.sa_handler) {
    g_prev.sa_handler(sig);
  }
  signal(SIGABRT, SIG_DFL);
  raise(SIGABRT);
}

void abort_trace_install(const char *path) {
  strncpy(g_path, path, sizeof g_path - 1);
  void *warm[4];
  (void)backtrace(warm, 4); /* loads libgcc now, not inside the handler */
  struct sigaction sa;
  memset(&sa, 0, sizeof sa);
  sa.sa_sigaction = on_abort;
  sa.sa_flags = SA_SIGINFO | SA_NODEFER;
  sigemptyset(&sa.sa_mask);
  sigaction(SIGABRT, &sa, &g_prev);
}
