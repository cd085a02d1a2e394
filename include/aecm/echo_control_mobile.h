/*
 * Forwarder so that callers written against the reference tree compile unchanged: the reference's only
 * caller includes "aecm/echo_control_mobile.h" (reference main.cc:18).  Put this repository's include/ on
 * the include path and that line finds the drop-in declarations (../echo_control_mobile.h).
 */
#include "../echo_control_mobile.h"
