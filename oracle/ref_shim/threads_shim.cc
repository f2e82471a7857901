/* oracle/ref_shim/threads_shim.cc -- TEST INFRASTRUCTURE ONLY.
 * The reference's ThreadPool starts std::thread::hardware_concurrency() workers (threadpool.cc:58-63).  On a box whose
 * container has a CPU quota far below the visible core count (256 visible, 16 usable) that is 16 x oversubscribed and
 * understates the reference.  This definition -- linked into oracle/_ref only, bound locally with -Bsymbolic-functions --
 * lets the measuring side choose the worker count through AWM_REF_THREADS (unset or 0: the machine's count, as upstream).
 * No reference source is changed. */
#include <thread>
#include <cstdlib>
#include <unistd.h>

unsigned int
std::thread::hardware_concurrency() noexcept
{
  if (const char *e = getenv ("AWM_REF_THREADS"))
    {
      const long v = strtol (e, nullptr, 10);
      if (v > 0)
        return (unsigned int) v;
    }
  const long n = sysconf (_SC_NPROCESSORS_ONLN);
  return n > 0 ? (unsigned int) n : 1u;
}
