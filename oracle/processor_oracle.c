/* placeholder so the Makefile links; filled in by the processor restatement */
int orx_processor_oracle_placeholder(void) { return 0; }
