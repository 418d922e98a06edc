/* Test infrastructure.  LD_PRELOAD for tests/wave_emul/stream_races.py: free() does nothing, so that no address is ever handed out twice --
 * the GPU's caching allocator keeps a stream's blocks to that stream; malloc would hand a side stream's freed buffer to the main stream
 * and the race detector would see two unrelated tensors as one. */
void free(void* p) { (void) p; }
