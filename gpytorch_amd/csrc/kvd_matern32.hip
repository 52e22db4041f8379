#define GPAMD_KIND gpamd::KIND_MATERN32
#define GPAMD_NAME matern32
#include "kvd_family.inc"
