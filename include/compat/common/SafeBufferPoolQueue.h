#include <sdrx/SafeBufferPoolQueue.h>
