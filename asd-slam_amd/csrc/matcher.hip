#include "ctx.h"
void matcher_free(asd_ctx*) {}
