#define GPAMD_KIND gpamd::KIND_MATERN32
#define GPAMD_NAME matern32
#include "kvh_family.inc"
