#define GPAMD_KIND gpamd::KIND_MATERN12
#define GPAMD_NAME matern12
#define GPAMD_NO_GRAM  // k = exp(-sqrt(s)) is not Lipschitz in s at 0: always the direct-difference kernel
#include "kv_family.inc"
