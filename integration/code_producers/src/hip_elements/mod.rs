// code_producers/src/hip_elements/mod.rs  --  NEW FILE of the `--hip` target.
//
// STATUS: written against circom v2.2.x's crates, NOT COMPILED in the environment this repository was built in (no
// cargo / rustc there).  The executable specification of everything below is the Python package `circom_amd`
// (`frontend/flatten.py` = the trace, `cwf.py` = this file's byte layout, `hip_backend.py` = the process `finish` runs);
// tests/test_cwf_backend.py pins the byte layout on the Python side, tests/test_integration_sources.py checks that the
// field lists and opcode tables of this file and of `circom_amd/cwf.py` / `opcodes.py` name the same things in the same
// order.
//
// Role, next to the existing producers (c_elements/mod.rs:6-39 `CProducer`, wasm_elements `WASMProducer`): a `HipProducer`
// does not print code.  The compiler's `WriteHip` implementation (integration/compiler/src/hip_backend.rs) EXECUTES the IR
// once at compile time and records, through the methods below, the flat witness program over GLOBAL signal ids;
// `finish` writes it as `<name>.cwf` and runs the lowering (`python -m circom_amd.hip_backend`), which emits
// `<name>.cwt` (batched evaluation schedule + bit-plane program), `<name>.dat` (the reference's layout, byte for byte:
// c_code_generator.rs:818-865) and `<name>.r1cs` for the R1CS check kernel.

use num_bigint_dig::BigInt;
use std::fs::File;
use std::io::{BufWriter, Write};
use std::path::Path;
use std::process::Command;

/// Opcodes of the flat program = `circom_amd/opcodes.py` (same numbers).  The first 24 mirror `OperatorType`
/// (compiler/src/intermediate_representation/compute_bucket.rs:7-34); address arithmetic never reaches the flat program
/// (indices are compile-time values of the trace).
#[repr(i64)]
#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub enum FlatOpcode {
    Copy = 0,
    Add = 1,
    Sub = 2,
    Mul = 3,
    Div = 4,
    IntDiv = 5,
    Mod = 6,
    Pow = 7,
    Neg = 8,
    ShiftL = 9,
    ShiftR = 10,
    BitAnd = 11,
    BitOr = 12,
    BitXor = 13,
    Complement = 14,
    Lesser = 15,
    Greater = 16,
    LesserEq = 17,
    GreaterEq = 18,
    Eq = 19,
    NotEq = 20,
    BoolAnd = 21,
    BoolOr = 22,
    BoolNot = 23,
    Select = 24,
    AssertEq = 25,
    AssertNz = 26,
    Run = 27,
    Call = 28,
    Log = 29,
}

/// Operand kinds of the flat program (`opcodes.py` K_*).
pub const K_SIG: i64 = 0;
pub const K_TMP: i64 = 1;
pub const K_CONST: i64 = 2;
pub const K_NONE: i64 = 3;

#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub struct Operand {
    pub kind: i64,
    pub value: i64,
}
impl Operand {
    pub fn signal(id: usize) -> Operand { Operand { kind: K_SIG, value: id as i64 } }
    pub fn temp(id: usize) -> Operand { Operand { kind: K_TMP, value: id as i64 } }
    pub fn constant(id: usize) -> Operand { Operand { kind: K_CONST, value: id as i64 } }
    pub fn none() -> Operand { Operand { kind: K_NONE, value: 0 } }
}

#[derive(Clone, Debug)]
pub struct FlatOp {
    pub op: FlatOpcode,
    pub dst: Operand,
    pub a: Operand,
    pub b: Operand,
    pub c: Operand,
}

/// One constraint A * B - C = 0 over global signal ids (signal 0 = the constant 1), coefficients canonical in [0, p):
/// what `constraint_list` already holds for the r1cs writer (constraint_writers/src/r1cs_writer.rs:49-91).
#[derive(Clone, Debug, Default)]
pub struct FlatConstraint {
    pub a: Vec<(u32, BigInt)>,
    pub b: Vec<(u32, BigInt)>,
    pub c: Vec<(u32, BigInt)>,
}

/// `IODef` of the Mixed component clusters (compiler/src/circuit_design/build.rs:488-520), as the `.cwf` carries it.
#[derive(Clone, Debug)]
pub struct FlatIoDef {
    pub offset: u32,
    pub dims: Vec<u32>,
    pub size: u32,
    pub bus_id: u32,
}

/// The ten u32 of the `.cwf` header, in file order (cwf.py: `struct.pack("<10I", ...)`).
pub const CWF_HEADER_FIELDS: [&str; 10] = [
    "n_signals",
    "n_temps",
    "n_constants",
    "main_input_start",
    "n_main_inputs",
    "n_public_inputs",
    "n_outputs",
    "n_input_names",
    "n_ops",
    "n_constraints",
];
pub const CWF_MAGIC: &[u8; 4] = b"CWFL";
pub const CWF_VERSION: u32 = 2;
/// Columns of the flat code section, in file order (cwf.py `COLS`), each n_ops x i64.
pub const CWF_CODE_COLUMNS: [&str; 9] = ["op", "dk", "dv", "ak", "av", "bk", "bv", "ck", "cv"];

#[derive(Default)]
pub struct HipProducer {
    pub prime: BigInt,
    pub prime_str: String,
    pub n64: usize,
    // numbering, exactly as CProducer holds it (c_elements/mod.rs:6-39): filled by compiler/src/circuit_design/build.rs
    pub total_number_of_signals: usize,
    pub main_signal_offset: usize,       // get_main_input_signal_start(): 1 + #outputs of main (c_elements/mod.rs:156-158)
    pub number_of_main_inputs: usize,
    pub number_of_main_outputs: usize,
    pub number_of_public_inputs: usize,
    pub main_input_list: Vec<(String, usize, usize)>,    // (name, first signal, size), InputList of c_elements/mod.rs:12
    pub field_tracking: Vec<String>,     // constant list (decimal strings), index = ValueBucket.value for BigInt values
    pub io_map: Vec<(u32, Vec<FlatIoDef>)>,
    pub string_table: Vec<String>,       // the producer's string table (c_elements/mod.rs get_string_table), set by build.rs
    pub log_strings: Vec<String>,        // the strings the trace's LOG rows name, in first-use order
    // the trace
    pub ops: Vec<FlatOp>,
    pub n_temps: usize,
    pub constraints: Vec<FlatConstraint>,
}

impl HipProducer {
    pub fn new_temp(&mut self) -> Operand {
        self.n_temps += 1;
        Operand::temp(self.n_temps - 1)
    }
    /// index in `log_strings` of string `id` of the producer's string table (LogBucketArg::LogStr)
    pub fn log_string(&mut self, id: usize) -> usize {
        let text = self.string_table[id].clone();
        match self.log_strings.iter().position(|t| *t == text) {
            Some(i) => i,
            None => {
                self.log_strings.push(text);
                self.log_strings.len() - 1
            }
        }
    }
    pub fn emit(&mut self, op: FlatOpcode, dst: Operand, a: Operand, b: Operand) {
        self.ops.push(FlatOp { op, dst, a, b, c: Operand::none() });
    }
    pub fn emit3(&mut self, op: FlatOpcode, dst: Operand, a: Operand, b: Operand, c: Operand) {
        self.ops.push(FlatOp { op, dst, a, b, c });
    }

    /// The constraints, read back from the `.r1cs` file circom has already written (`--hip` implies `--r1cs`,
    /// circom/src/execution_user.rs:27-57): `compiler::Circuit` holds none (circuit_design/circuit.rs:17-20; they live in
    /// `constraint_list`, which the `compiler` and `code_producers` crates do not depend on).  Layout as
    /// constraint_writers/src/r1cs_writer.rs writes it: "r1cs" | version | n sections | sections {type u32, size u64}:
    /// 1 = header (field size u32, prime, nWires u32, nPubOut, nPubIn, nPrvIn u32, nLabels u64, mConstraints u32),
    /// 2 = constraints (3 x {n u32, n x {wire u32, coefficient}}), 3 = wire -> label (u64 each; label = signal id).
    pub fn load_r1cs(&mut self, path: &Path) -> std::io::Result<()> {
        use std::io::{Error, ErrorKind};
        let bad = |m: &str| Error::new(ErrorKind::InvalidData, m.to_string());
        let buf = std::fs::read(path)?;
        let u32_at = |o: usize| -> std::io::Result<u32> {
            buf.get(o..o + 4).map(|b| u32::from_le_bytes([b[0], b[1], b[2], b[3]])).ok_or_else(|| bad("r1cs truncated"))
        };
        let u64_at = |o: usize| -> std::io::Result<u64> {
            buf.get(o..o + 8).map(|b| u64::from_le_bytes([b[0], b[1], b[2], b[3], b[4], b[5], b[6], b[7]])).ok_or_else(|| bad("r1cs truncated"))
        };
        if buf.len() < 12 || &buf[0..4] != b"r1cs" {
            return Err(bad("not an r1cs file"));
        }
        let n_sections = u32_at(8)? as usize;
        let mut at = 12usize;
        let (mut hdr, mut cons, mut map) = (None, None, None);
        for _ in 0..n_sections {
            let (ty, size) = (u32_at(at)?, u64_at(at + 4)? as usize);
            at += 12;
            match ty {
                1 => hdr = Some(at),
                2 => cons = Some(at),
                3 => map = Some((at, size / 8)),
                _ => {}
            }
            at += size;
        }
        let (hdr, cons, (map_at, n_wires)) = match (hdr, cons, map) {
            (Some(h), Some(c), Some(m)) => (h, c, m),
            _ => return Err(bad("r1cs section missing")),
        };
        let fs = u32_at(hdr)? as usize;
        let m_constraints = u32_at(hdr + 4 + fs + 4 + 4 + 4 + 4 + 8)? as usize;
        let mut label = Vec::with_capacity(n_wires);
        for w in 0..n_wires {
            label.push(u64_at(map_at + 8 * w)? as u32);
        }
        let mut o = cons;
        self.constraints.clear();
        for _ in 0..m_constraints {
            let mut parts: [Vec<(u32, BigInt)>; 3] = [Vec::new(), Vec::new(), Vec::new()];
            for part in parts.iter_mut() {
                let n = u32_at(o)? as usize;
                o += 4;
                for _ in 0..n {
                    let wire = u32_at(o)? as usize;
                    let coef = buf.get(o + 4..o + 4 + fs).ok_or_else(|| bad("r1cs truncated"))?;
                    part.push((*label.get(wire).ok_or_else(|| bad("r1cs wire out of range"))?, BigInt::from_bytes_le(num_bigint_dig::Sign::Plus, coef)));
                    o += 4 + fs;
                }
            }
            let [a, b, c] = parts;
            self.constraints.push(FlatConstraint { a, b, c });
        }
        Ok(())
    }

    fn le_bytes(&self, v: &BigInt) -> Vec<u8> {
        let (_, mut bytes) = v.to_bytes_le();
        bytes.resize(8 * self.n64, 0);
        bytes
    }

    /// `<name>.cwf`, byte for byte what `circom_amd/cwf.py::write_cwf` writes.
    pub fn write_cwf(&self, path: &Path) -> std::io::Result<()> {
        let mut w = BufWriter::new(File::create(path)?);
        w.write_all(CWF_MAGIC)?;
        for v in [CWF_VERSION, self.n64 as u32, 0u32] {
            w.write_all(&v.to_le_bytes())?;
        }
        w.write_all(&self.le_bytes(&self.prime))?;
        let header: [u32; 10] = [
            self.total_number_of_signals as u32,
            self.n_temps as u32,
            self.field_tracking.len() as u32,
            self.main_signal_offset as u32,
            self.number_of_main_inputs as u32,
            self.number_of_public_inputs as u32,
            self.number_of_main_outputs as u32,
            self.main_input_list.len() as u32,
            self.ops.len() as u32,
            self.constraints.len() as u32,
        ];
        for v in header {
            w.write_all(&v.to_le_bytes())?;
        }
        for c in &self.field_tracking {
            let v = BigInt::parse_bytes(c.as_bytes(), 10).expect("constant list holds decimal strings");
            w.write_all(&self.le_bytes(&v))?;
        }
        for (name, start, size) in &self.main_input_list {
            w.write_all(&(name.len() as u32).to_le_bytes())?;
            w.write_all(name.as_bytes())?;
            w.write_all(&(*start as u32).to_le_bytes())?;
            w.write_all(&(*size as u32).to_le_bytes())?;
        }
        // nine columns, each n_ops x i64: op, dk, dv, ak, av, bk, bv, ck, cv
        let col = |f: &dyn Fn(&FlatOp) -> i64, w: &mut BufWriter<File>| -> std::io::Result<()> {
            for o in &self.ops {
                w.write_all(&f(o).to_le_bytes())?;
            }
            Ok(())
        };
        col(&|o| o.op as i64, &mut w)?;
        col(&|o| o.dst.kind, &mut w)?;
        col(&|o| o.dst.value, &mut w)?;
        col(&|o| o.a.kind, &mut w)?;
        col(&|o| o.a.value, &mut w)?;
        col(&|o| o.b.kind, &mut w)?;
        col(&|o| o.b.value, &mut w)?;
        col(&|o| o.c.kind, &mut w)?;
        col(&|o| o.c.value, &mut w)?;
        for cons in &self.constraints {
            for part in [&cons.a, &cons.b, &cons.c] {
                let mut terms = part.clone();
                terms.sort_by_key(|t| t.0);                    // cwf.py writes the terms of a part by signal id
                w.write_all(&(terms.len() as u32).to_le_bytes())?;
                for (sig, coef) in terms {
                    w.write_all(&sig.to_le_bytes())?;
                    w.write_all(&self.le_bytes(&coef))?;
                }
            }
        }
        w.write_all(&0u32.to_le_bytes())?;                     // functions: CallBuckets are inlined by the trace or refused (hip_backend.rs)
        w.write_all(&(self.io_map.len() as u32).to_le_bytes())?;
        for (tid, defs) in &self.io_map {
            w.write_all(&tid.to_le_bytes())?;
            w.write_all(&(defs.len() as u32).to_le_bytes())?;
            for d in defs {
                w.write_all(&d.offset.to_le_bytes())?;
                w.write_all(&(d.dims.len() as u32).to_le_bytes())?;
                for l in &d.dims {
                    w.write_all(&l.to_le_bytes())?;
                }
                w.write_all(&d.size.to_le_bytes())?;
                w.write_all(&d.bus_id.to_le_bytes())?;
            }
        }
        w.write_all(&(self.log_strings.len() as u32).to_le_bytes())?;
        for t in &self.log_strings {
            w.write_all(&(t.len() as u32).to_le_bytes())?;
            w.write_all(t.as_bytes())?;
        }
        w.flush()
    }

    /// Writes `<dir>/<name>.cwf` and runs the lowering, the way the `--c` target leaves `make` to the user but in one
    /// step: the artefacts the runtime loads (`cw_load`) are `<name>.cwt`, `<name>.dat`, `<name>.r1cs` in `dir`.
    pub fn finish(&self, dir: &Path, name: &str) -> Result<(), ()> {
        let cwf = dir.join(format!("{}.cwf", name));
        self.write_cwf(&cwf).map_err(|_| {})?;
        let status = Command::new(std::env::var("CIRCOM_HIP_PYTHON").unwrap_or_else(|_| "python3".to_string()))
            .args(["-m", "circom_amd.hip_backend"])
            .arg(&cwf)
            .arg("-o")
            .arg(dir)
            .status()
            .map_err(|_| {})?;
        if status.success() { Ok(()) } else { Err(()) }
    }
}
