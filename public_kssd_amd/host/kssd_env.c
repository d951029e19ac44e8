/* kssd_env.c -- libkssd_env.so: the command's OpenMP wait policy, set before libgomp reads its environment.
 *
 * The host threads of `kssd` wait for one another every few milliseconds (a wave of input files, a job's genomes), and libgomp's
 * threads wait by SPINNING unless OMP_WAIT_POLICY says otherwise at the moment the library initialises -- there is no call for it.
 * Spinning teams are what a container's CPU quota throttles: on the GPU hosts (16 CPUs) the command lost two accounting periods per
 * run to it.  The loader initialises the libraries of a program in the reverse of the order it loaded them
 * (dependencies first): the command line links this library BEHIND libgomp, so this constructor runs in front of libgomp's.  Whether
 * it did can be asked through a second variable whose effect is visible (the thread limit); where it did not, the command runs on
 * with spinning teams and says so under KSSD_TIMING (host/kssd_cli.c) -- it is never restarted. */
#include <stdlib.h>

#define KSSD_ENV_THREAD_LIMIT "1000003" /* (a limit nobody reaches: the mark main() looks for with omp_get_thread_limit()) */

__attribute__((constructor)) static void kssd_env_set(void)
{
    if (getenv("OMP_WAIT_POLICY") || getenv("GOMP_SPINCOUNT")) return; /* the caller's choice stands */
    setenv("OMP_WAIT_POLICY", "passive", 0);
    setenv("OMP_THREAD_LIMIT", KSSD_ENV_THREAD_LIMIT, 0);
}
