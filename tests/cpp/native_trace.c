/* native_trace.c -- test aid: on SIGSEGV / SIGBUS / SIGABRT writes the native call stack of the faulting thread to stderr (module +
 * offset per frame, resolvable with addr2line / llvm-symbolizer), then lets the previously installed handler (Python's faulthandler)
 * run.  Loaded by tests/sharded_rank_process.py so that a crash inside a native library names its place. */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <string.h>
#include <unistd.h>

static struct sigaction g_old[64];

static void on_fault(int sig, siginfo_t* info, void* uctx)
{
	static const char head[] = "native_trace: call stack of the faulting thread\n";
	void* frames[64];
	(void)!write(2, head, sizeof head - 1);
	const int n = backtrace(frames, 64);
	backtrace_symbols_fd(frames, n, 2);
	if (sig > 0 && sig < 64) {
		const struct sigaction* old = &g_old[sig];
		if ((old->sa_flags & SA_SIGINFO) && old->sa_sigaction) { old->sa_sigaction(sig, info, uctx); return; }
		if (old->sa_handler && old->sa_handler != SIG_DFL && old->sa_handler != SIG_IGN) { old->sa_handler(sig); return; }
	}
	signal(sig, SIG_DFL);
	raise(sig);
}

/* call after faulthandler.enable(): the handlers installed before are chained to */
int native_trace_install(void)
{
	static const int sigs[3] = {SIGSEGV, SIGBUS, SIGABRT};
	void* warm[2];
	backtrace(warm, 2);      /* (loads libgcc now, not inside the handler) */
	for (int i = 0; i < 3; ++i) {
		struct sigaction sa;
		memset(&sa, 0, sizeof sa);
		sa.sa_sigaction = on_fault;
		sa.sa_flags = SA_SIGINFO | SA_ONSTACK | SA_NODEFER;
		sigemptyset(&sa.sa_mask);
		if (sigaction(sigs[i], &sa, &g_old[sigs[i]]) != 0) return -1;
	}
	return 0;
}
