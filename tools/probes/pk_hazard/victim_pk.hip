#include <hip/hip_runtime.h>
#define VICTIM_NAME(x) pk_##x
#include "victim.inc"
