#include <hip/hip_runtime.h>
#define VICTIM_NAME(x) nopk_##x
#include "victim.inc"
