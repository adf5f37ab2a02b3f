/* ORACLE — TEST INFRASTRUCTURE ONLY: how many OpenMP threads the CPU baseline used. */
#include <omp.h>
int orc_num_threads(void) { return omp_get_max_threads(); }
