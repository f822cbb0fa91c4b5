#include "ctx.h"
