#define GPAMD_KIND gpamd::KIND_RQ
#define GPAMD_NAME rq
#include "kvd_family.inc"
