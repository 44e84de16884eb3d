/* TEST INFRASTRUCTURE (tests/emu): the fibre switch of the lock-step interpreter -- System V x86-64, callee-saved registers only. */
    .text
    .globl hipemu_switch
    .type hipemu_switch, @function
hipemu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
    .size hipemu_switch, .-hipemu_switch
    .section .note.GNU-stack,"",@progbits
