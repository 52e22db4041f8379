#define GPAMD_KIND gpamd::KIND_MATERN12
#define GPAMD_NAME matern12
#include "kvd_family.inc"
