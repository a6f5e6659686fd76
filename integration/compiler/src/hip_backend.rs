// compiler/src/hip_backend.rs  --  NEW FILE of the `--hip` target: `WriteHip`, the third translating trait
// beside `WriteC` / `WriteWasm` (compiler/src/translating_traits/mod.rs:5-29).
//
// STATUS: NOT COMPILED in the environment this repository was built in (no cargo).  Executable specification:
// `circom_amd/frontend/dsl.py` + `flatten.py` (the same trace over a Python eDSL), pinned against the reference runtime by
// the repository's tests.  What this file does differently from `produce_c`: it does not PRINT the witness program, it
// RUNS its control flow once at compile time -
//   * CreateCmpBucket (create_component_bucket.rs:7-31)  allocates the component: global signal start = father's start +
//     signal_offset (+ jump * index), component id = father's id + component_offset (+ jump * index): exactly the
//     arguments `<T>_<id>_create` receives (template.rs:177-279);
//   * LoopBucket / BranchBucket whose condition is a compile-time value are unrolled / resolved (they always are inside
//     templates when they enclose signal declarations or constraints: `ConstraintInUnknown`, execute.rs:147-156);
//   * StoreBucket to a sub-component input decrements its input counter and, when it reaches zero, traces the child's body
//     on the spot (store_bucket.rs:660-735: "need to run sub component"); a component without inputs is traced at creation
//     (template.rs:274-278);
//   * ValueBucket / LoadBucket / ComputeBucket are evaluated when every operand is a compile-time value (loop counters,
//     template parameters: what `ir_processing/reduce_stack.rs:28-50` folds for addresses) and otherwise RECORDED as flat
//     operations over global signal ids through `HipProducer::emit`; `lvar`s that depend on signals become temporaries;
//   * LocationRule::Mapped (store_bucket.rs:498-566) needs no run-time table: the component a store goes to is known
//     while tracing, so `io_map[template_id].defs[signal_code]` is read at compile time;
//   * AssertBucket (assert_bucket.rs:7-12) becomes ASSERT_NZ / ASSERT_EQ;
//   * CallBucket of a function whose control flow is known is traced inline; data-dependent loops / branches / addresses
//     (`Fr_isTrue` / `Fr_toInt` on signal values: loop_bucket.rs:76-91, branch_bucket.rs:100-122,
//     compute_bucket.rs:361-363) are refused with `HipError::RunTimeControl` - the Python front-end lowers those to the
//     register bytecode of `circom_amd/frontend/rtcode.py` (a `CALL` row); porting that translation is the next step of
//     this file and is marked TODO(tier2) below.

use crate::circuit_design::circuit::Circuit;
use crate::circuit_design::template::TemplateCode;
use crate::intermediate_representation::ir_interface::*;
use code_producers::hip_elements::{FlatOpcode, HipProducer, Operand, K_NONE};
use num_bigint_dig::BigInt;
use circom_algebra::modular_arithmetic as ma;

#[derive(Debug)]
pub enum HipError {
    RunTimeControl { line: usize, what: &'static str },
    Unsupported { line: usize, what: &'static str },
    Internal(&'static str),
}

/// A value during the trace: known now, or living in a signal / temporary / constant-table slot at run time.
#[derive(Clone, Debug)]
pub enum Val {
    Known(BigInt),
    Run(Operand),
}

pub struct CmpState {
    pub template_id: usize,
    pub signal_start: usize,
    pub input_counter: usize,
    pub subcomponents: Vec<usize>,          // component ids, indexed like mySubcomponents[]
    pub ran: bool,
}

pub struct Tracer<'a> {
    pub circuit: &'a Circuit,
    pub p: &'a mut HipProducer,
    pub components: Vec<Option<CmpState>>,  // ctx->componentMemory[]
}

/// One activation of a template body (`<T>_<id>_run`): its lvar arena and expaux are `Val`s.
struct Frame {
    cmp: usize,                              // ctx_index
    lvar: Vec<Val>,
}

pub trait WriteHip {
    fn produce_hip(&self, producer: &mut HipProducer) -> Result<(), HipError>;
}

impl WriteHip for Circuit {
    /// `run(ctx)` (circuit.rs:526-561): create main at signal 1 / component 0, run it if it has inputs.
    fn produce_hip(&self, producer: &mut HipProducer) -> Result<(), HipError> {
        let n_cmp = self.c_producer.number_of_components;
        let mut t = Tracer { circuit: self, p: producer, components: (0..n_cmp).map(|_| None).collect() };
        // main's TemplateCode is the one whose header is CProducer::get_main_header() (c_elements/mod.rs:141; set by
        // build.rs:259-298 from the main component's template instance, "Main_0" style)
        let main_header = self.c_producer.get_main_header();
        let main_id = self.templates.iter().position(|t| t.header == main_header)
            .ok_or(HipError::Internal("main template not found among the circuit's templates"))?;
        t.create(main_id, 1, 0)?;
        if t.components[0].as_ref().unwrap().input_counter > 0 {
            // main's inputs are the circuit's inputs: they are set before run(), so the body runs now
            t.run_component(0)?;
        }
        Ok(())
    }
}

impl<'a> Tracer<'a> {
    fn template(&self, id: usize) -> &'a TemplateCode {
        &self.circuit.templates[id]
    }

    /// `<T>_<id>_create` (template.rs:177-279)
    fn create(&mut self, template_id: usize, signal_start: usize, cmp: usize) -> Result<(), HipError> {
        let tc = self.template(template_id);
        self.components[cmp] = Some(CmpState {
            template_id,
            signal_start,
            input_counter: tc.number_of_inputs,
            subcomponents: vec![0; tc.number_of_components],
            ran: false,
        });
        if tc.number_of_inputs == 0 {
            self.run_component(cmp)?;                           // template.rs:274-278
        }
        Ok(())
    }

    /// `<T>_<id>_run` (template.rs:281-472)
    fn run_component(&mut self, cmp: usize) -> Result<(), HipError> {
        let template_id = self.components[cmp].as_ref().unwrap().template_id;
        let tc = self.template(template_id);
        self.components[cmp].as_mut().unwrap().ran = true;
        let mut f = Frame { cmp, lvar: vec![Val::Known(BigInt::from(0)); tc.var_stack_depth] };
        self.exec_list(&tc.body, &mut f)
    }

    fn exec_list(&mut self, list: &InstructionList, f: &mut Frame) -> Result<(), HipError> {
        for i in list {
            self.exec(i, f)?;
        }
        Ok(())
    }

    fn exec(&mut self, i: &Instruction, f: &mut Frame) -> Result<(), HipError> {
        match i {
            Instruction::CreateCmp(b) => {
                // create_component_bucket.rs:219-347: defined positions of a (possibly multi-dimensional, possibly mixed) array
                let first = self.known_usize(&b.sub_cmp_id, f, b.line)?;
                let me = self.components[f.cmp].as_ref().unwrap();
                let (my_start, my_id) = (me.signal_start, f.cmp);
                for (k, (pos, _parallel)) in b.defined_positions.iter().enumerate() {
                    let _ = k;
                    let cmp_id = my_id + b.component_offset + pos * b.component_offset_jump;
                    let sig = my_start + b.signal_offset + pos * b.signal_offset_jump;
                    self.components[my_id].as_mut().unwrap().subcomponents[first + pos] = cmp_id;
                    self.create(b.template_id, sig, cmp_id)?;
                }
                Ok(())
            }
            Instruction::Loop(b) => {
                // loop_bucket.rs:76-91: `while (Fr_isTrue(cond))`
                loop {
                    match self.eval(&b.continue_condition, f)? {
                        Val::Known(c) => {
                            if !ma::as_bool(&c, &self.p.prime) {
                                return Ok(());
                            }
                        }
                        Val::Run(_) => return Err(HipError::RunTimeControl { line: b.line, what: "loop on a signal value" }),   // TODO(tier2)
                    }
                    self.exec_list(&b.body, f)?;
                }
            }
            Instruction::Branch(b) => match self.eval(&b.cond, f)? {
                Val::Known(c) => {
                    if ma::as_bool(&c, &self.p.prime) { self.exec_list(&b.if_branch, f) } else { self.exec_list(&b.else_branch, f) }
                }
                Val::Run(cond) => self.exec_predicated(b, cond, f),
            },
            Instruction::Store(b) => self.exec_store(b, f),
            Instruction::Assert(b) => {
                match self.eval(&b.evaluate, f)? {
                    Val::Known(c) => {
                        if !ma::as_bool(&c, &self.p.prime) {
                            return Err(HipError::Unsupported { line: b.line, what: "assert is false at compile time" });
                        }
                    }
                    Val::Run(x) => self.p.emit(FlatOpcode::AssertNz, Operand::none(), x, Operand::none()),
                }
                Ok(())
            }
            Instruction::Log(b) => {
                // LogBucket (log_bucket.rs:105-162): one LOG row per argument - the value as operand a, or kind NONE with the
                // string-table index (-1: `log()` without arguments) - and dst.value = 1 on the row that ends the statement.
                // The lowering keeps every logged value in the table; cw_get_log formats the reference's stdout per instance.
                let n = b.argsprint.len();
                if n == 0 {
                    self.p.emit(FlatOpcode::Log, Operand { kind: K_NONE, value: 1 }, Operand { kind: K_NONE, value: -1 }, Operand::none());
                }
                for (k, arg) in b.argsprint.iter().enumerate() {
                    let last = if k + 1 == n { 1 } else { 0 };
                    let a = match arg {
                        LogBucketArg::LogExp(e) => { let v = self.eval(e, f)?; self.materialise(v) }
                        LogBucketArg::LogStr(id) => Operand { kind: K_NONE, value: self.p.log_string(*id) as i64 },
                    };
                    self.p.emit(FlatOpcode::Log, Operand { kind: K_NONE, value: last }, a, Operand::none());
                }
                Ok(())
            }
            Instruction::Call(b) => Err(HipError::RunTimeControl { line: b.line, what: "function call as a statement" }),   // TODO(tier2)
            Instruction::Return(b) => Err(HipError::Unsupported { line: b.line, what: "return outside a traced function" }),
            Instruction::Value(_) | Instruction::Load(_) | Instruction::Compute(_) => {
                self.eval(i, f).map(|_| ())
            }
        }
    }

    /// A value-dependent `if` inside `<--` code (sugar of the ternary operator): both sides are evaluated and every store is
    /// predicated: dst = cond ? new : old (SELECT), as `frontend/dsl.py::select` does.
    fn exec_predicated(&mut self, b: &BranchBucket, cond: Operand, f: &mut Frame) -> Result<(), HipError> {
        for (side, list) in [(true, &b.if_branch), (false, &b.else_branch)] {
            for ins in list {
                let st = match &**ins {
                    Instruction::Store(st) => st,
                    _ => return Err(HipError::RunTimeControl { line: b.line, what: "statement other than a store under a signal-dependent condition" }),
                };
                let new = self.eval(&st.src, f)?;
                let new = self.materialise(new);
                let (dst, is_var) = self.store_target(st, f)?;
                let old = if is_var { self.materialise(f.lvar[dst.value as usize].clone()) } else { dst };
                let t = self.p.new_temp();
                let (x, y) = if side { (new, old) } else { (old, new) };
                self.p.emit3(FlatOpcode::Select, t, cond, x, y);
                if is_var {
                    f.lvar[dst.value as usize] = Val::Run(t);
                } else {
                    self.p.emit(FlatOpcode::Copy, dst, t, Operand::none());
                }
            }
        }
        Ok(())
    }

    /// Destination of a store: (operand, is_lvar).  Signals are GLOBAL ids (component's signal start + local offset).
    fn store_target(&mut self, b: &StoreBucket, f: &mut Frame) -> Result<(Operand, bool), HipError> {
        match &b.dest_address_type {
            AddressType::Variable => {
                let idx = self.location(&b.dest, None, f, b.line)?;
                Ok((Operand { kind: -1, value: idx as i64 }, true))
            }
            AddressType::Signal => {
                let idx = self.location(&b.dest, None, f, b.line)?;
                let start = self.components[f.cmp].as_ref().unwrap().signal_start;
                Ok((Operand::signal(start + idx), false))
            }
            AddressType::SubcmpSignal { cmp_address, .. } => {
                let k = self.known_usize(cmp_address, f, b.line)?;
                let child = self.components[f.cmp].as_ref().unwrap().subcomponents[k];
                let idx = self.location(&b.dest, Some(child), f, b.line)?;
                let start = self.components[child].as_ref().unwrap().signal_start;
                Ok((Operand::signal(start + idx), false))
            }
        }
    }

    fn exec_store(&mut self, b: &StoreBucket, f: &mut Frame) -> Result<(), HipError> {
        let n = b.context.size.known_size();                      // multi-element copies: SizeOption::Single (types.rs:18-21)
        let src0 = self.eval(&b.src, f)?;
        let (dst, is_var) = self.store_target(b, f)?;
        for e in 0..n {
            let src = if e == 0 { src0.clone() } else { self.eval_offset(&b.src, e, f)? };
            if is_var {
                f.lvar[dst.value as usize + e] = src;             // stays a compile-time value when it is one
            } else {
                let s = self.materialise(src);
                self.p.emit(FlatOpcode::Copy, Operand::signal(dst.value as usize + e), s, Operand::none());
            }
        }
        // store_bucket.rs:660-735: an input of a sub-component
        if let AddressType::SubcmpSignal { cmp_address, input_information: InputInformation::Input { .. }, .. } = &b.dest_address_type {
            let k = self.known_usize(cmp_address, f, b.line)?;
            let child = self.components[f.cmp].as_ref().unwrap().subcomponents[k];
            let st = self.components[child].as_mut().unwrap();
            st.input_counter -= n;
            if st.input_counter == 0 && !st.ran {
                self.run_component(child)?;
            }
        }
        Ok(())
    }

    /// LocationRule -> local offset (location_rule.rs:33-36).  `Indexed` locations are address computations over
    /// compile-time values; `Mapped` ones are resolved through the io map of the (known) component's template.
    fn location(&mut self, rule: &LocationRule, child: Option<usize>, f: &mut Frame, line: usize) -> Result<usize, HipError> {
        match rule {
            LocationRule::Indexed { location, .. } => self.known_usize(location, f, line),
            LocationRule::Mapped { signal_code, indexes } => {
                let child = child.ok_or(HipError::Unsupported { line, what: "Mapped location outside a sub-component access" })?;
                let tid = self.components[child].as_ref().unwrap().template_id as u32;
                let defs = &self.p.io_map.iter().find(|(t, _)| *t == tid).ok_or(HipError::Unsupported { line, what: "template missing from the io map" })?.1;
                let def = defs[*signal_code].clone();
                let mut off = def.offset as usize;
                for a in indexes {
                    match a {
                        AccessType::Indexed(info) => {
                            // ((i0 * len1 + i1) * len2 + ...) * size, store_bucket.rs:525-552
                            let mut idx = 0usize;
                            for (j, ip) in info.indexes.iter().enumerate() {
                                let v = self.known_usize(ip, f, line)?;
                                idx = if j == 0 { v } else { idx * def.dims[j] as usize + v };
                            }
                            for j in info.indexes.len()..info.symbol_dim {
                                idx *= def.dims[j] as usize;
                            }
                            off += idx * def.size as usize;
                        }
                        AccessType::Qualified(_) => return Err(HipError::Unsupported { line, what: "bus field access" }),
                    }
                }
                Ok(off)
            }
        }
    }

    fn known_usize(&mut self, ip: &InstructionPointer, f: &mut Frame, line: usize) -> Result<usize, HipError> {
        match self.eval(ip, f)? {
            Val::Known(v) => Ok(bigint_to_usize(&v)),
            Val::Run(_) => Err(HipError::RunTimeControl { line, what: "address that depends on a signal value (Fr_toInt)" }),   // TODO(tier2)
        }
    }

    /// Run-time operand of a value: a known value goes to the constant table (field_tracking).
    fn materialise(&mut self, v: Val) -> Operand {
        match v {
            Val::Run(o) => o,
            Val::Known(k) => {
                let s = k.to_str_radix(10);
                let id = match self.p.field_tracking.iter().position(|c| *c == s) {
                    Some(i) => i,
                    None => {
                        self.p.field_tracking.push(s);
                        self.p.field_tracking.len() - 1
                    }
                };
                Operand::constant(id)
            }
        }
    }

    fn eval_offset(&mut self, _ip: &InstructionPointer, _e: usize, _f: &mut Frame) -> Result<Val, HipError> {
        // element e of a multi-element source (load_bucket.rs produces a pointer, the store copies `size` elements): the
        // source location + e; left to the port of frontend/dsl.py's array stores
        Err(HipError::Unsupported { line: 0, what: "multi-element store" })
    }

    /// Expressions: ValueBucket (value_bucket.rs:75-90), LoadBucket (load_bucket.rs), ComputeBucket (compute_bucket.rs:315-341).
    fn eval(&mut self, i: &Instruction, f: &mut Frame) -> Result<Val, HipError> {
        match i {
            Instruction::Value(b) => Ok(match b.parse_as {
                ValueType::U32 => Val::Known(BigInt::from(b.value)),
                ValueType::BigInt => Val::Known(BigInt::parse_bytes(self.p.field_tracking[b.value].as_bytes(), 10).unwrap()),
            }),
            Instruction::Load(b) => match &b.address_type {
                AddressType::Variable => {
                    let idx = self.location(&b.src, None, f, b.line)?;
                    Ok(f.lvar[idx].clone())
                }
                AddressType::Signal => {
                    let idx = self.location(&b.src, None, f, b.line)?;
                    let start = self.components[f.cmp].as_ref().unwrap().signal_start;
                    Ok(Val::Run(Operand::signal(start + idx)))
                }
                AddressType::SubcmpSignal { cmp_address, .. } => {
                    let k = self.known_usize(cmp_address, f, b.line)?;
                    let child = self.components[f.cmp].as_ref().unwrap().subcomponents[k];
                    let idx = self.location(&b.src, Some(child), f, b.line)?;
                    let start = self.components[child].as_ref().unwrap().signal_start;
                    Ok(Val::Run(Operand::signal(start + idx)))
                }
            },
            Instruction::Compute(b) => {
                let mut args = Vec::with_capacity(b.stack.len());
                for a in &b.stack {
                    args.push(self.eval(a, f)?);
                }
                if args.iter().all(|a| matches!(a, Val::Known(_))) {
                    let k: Vec<&BigInt> = args.iter().map(|a| if let Val::Known(v) = a { v } else { unreachable!() }).collect();
                    return Ok(Val::Known(fold(&b.op, &k, &self.p.prime, b.line)?));
                }
                let (op, unary) = flat_opcode(&b.op, b.line)?;
                let a = self.materialise(args[0].clone());
                let bb = if unary { Operand::none() } else { self.materialise(args[1].clone()) };
                let t = self.p.new_temp();
                self.p.emit(op, t, a, bb);
                Ok(Val::Run(t))
            }
            Instruction::Call(b) => Err(HipError::RunTimeControl { line: b.line, what: "function call in an expression" }),   // TODO(tier2): trace inline when control is known
            _ => Err(HipError::Unsupported { line: 0, what: "statement in expression position" }),
        }
    }
}

/// OperatorType -> flat opcode (compute_bucket.rs:315-341 maps the same operators to Fr_* symbols).
fn flat_opcode(op: &OperatorType, line: usize) -> Result<(FlatOpcode, bool), HipError> {
    use OperatorType::*;
    Ok(match op {
        Mul => (FlatOpcode::Mul, false),
        Div => (FlatOpcode::Div, false),
        Add => (FlatOpcode::Add, false),
        Sub => (FlatOpcode::Sub, false),
        Pow => (FlatOpcode::Pow, false),
        IntDiv => (FlatOpcode::IntDiv, false),
        Mod => (FlatOpcode::Mod, false),
        ShiftL => (FlatOpcode::ShiftL, false),
        ShiftR => (FlatOpcode::ShiftR, false),
        LesserEq => (FlatOpcode::LesserEq, false),
        GreaterEq => (FlatOpcode::GreaterEq, false),
        Lesser => (FlatOpcode::Lesser, false),
        Greater => (FlatOpcode::Greater, false),
        Eq(_) => (FlatOpcode::Eq, false),
        NotEq => (FlatOpcode::NotEq, false),
        BoolOr => (FlatOpcode::BoolOr, false),
        BoolAnd => (FlatOpcode::BoolAnd, false),
        BitOr => (FlatOpcode::BitOr, false),
        BitAnd => (FlatOpcode::BitAnd, false),
        BitXor => (FlatOpcode::BitXor, false),
        PrefixSub => (FlatOpcode::Neg, true),
        BoolNot => (FlatOpcode::BoolNot, true),
        Complement => (FlatOpcode::Complement, true),
        ToAddress | MulAddress | AddAddress => {
            return Err(HipError::RunTimeControl { line, what: "address arithmetic on a signal value" })
        }
    })
}

/// Compile-time value of an operator on known operands: `circom_algebra::modular_arithmetic` (the semantic twin of the
/// run-time library), address operators on plain integers.
fn fold(op: &OperatorType, a: &[&BigInt], p: &BigInt, line: usize) -> Result<BigInt, HipError> {
    use OperatorType::*;
    let err = |what| HipError::Unsupported { line, what };
    Ok(match op {
        Mul => ma::mul(a[0], a[1], p),
        Div => ma::div(a[0], a[1], p).map_err(|_| err("division by zero at compile time"))?,
        Add => ma::add(a[0], a[1], p),
        Sub => ma::sub(a[0], a[1], p),
        Pow => ma::pow(a[0], a[1], p),
        IntDiv => ma::idiv(a[0], a[1], p).map_err(|_| err("division by zero at compile time"))?,
        Mod => ma::mod_op(a[0], a[1], p).map_err(|_| err("division by zero at compile time"))?,
        ShiftL => ma::shift_l(a[0], a[1], p).map_err(|_| err("shift amount"))?,
        ShiftR => ma::shift_r(a[0], a[1], p).map_err(|_| err("shift amount"))?,
        LesserEq => ma::lesser_eq(a[0], a[1], p),
        GreaterEq => ma::greater_eq(a[0], a[1], p),
        Lesser => ma::lesser(a[0], a[1], p),
        Greater => ma::greater(a[0], a[1], p),
        Eq(_) => ma::eq(a[0], a[1], p),
        NotEq => ma::not_eq(a[0], a[1], p),
        BoolOr => ma::bool_or(a[0], a[1], p),
        BoolAnd => ma::bool_and(a[0], a[1], p),
        BitOr => ma::bit_or(a[0], a[1], p),
        BitAnd => ma::bit_and(a[0], a[1], p),
        BitXor => ma::bit_xor(a[0], a[1], p),
        PrefixSub => ma::prefix_sub(a[0], p),
        BoolNot => ma::not(a[0], p),
        Complement => ma::complement(a[0], p),
        ToAddress => a[0].clone(),
        MulAddress => a[0] * a[1],
        AddAddress => a[0] + a[1],
    })
}

fn bigint_to_usize(v: &BigInt) -> usize {
    use num_traits::ToPrimitive;
    v.to_usize().expect("address fits usize")
}
