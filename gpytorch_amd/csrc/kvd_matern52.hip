#define GPAMD_KIND gpamd::KIND_MATERN52
#define GPAMD_NAME matern52
#include "kvd_family.inc"
