#include "ctx.h"
void ba_free(asd_ctx*) {}
