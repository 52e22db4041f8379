#define GPAMD_KIND gpamd::KIND_MATERN52
#define GPAMD_NAME matern52
#include "kvh_family.inc"
